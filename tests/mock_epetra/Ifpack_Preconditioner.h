// mock: the pure virtuals of Ifpack_Preconditioner (Trilinos packages/ifpack/src/Ifpack_Preconditioner.h)
#ifndef MOCK_IFPACK_PRECONDITIONER_H
#define MOCK_IFPACK_PRECONDITIONER_H
#include <ostream>
#include "Epetra_Operator.h"
#include "Epetra_RowMatrix.h"
#include "Teuchos_ParameterList.hpp"
enum Ifpack_CondestType { Ifpack_Cheap, Ifpack_CG, Ifpack_GMRES };
class Ifpack_Preconditioner : public Epetra_Operator {
 public:
  virtual int SetParameters(Teuchos::ParameterList& List) = 0;
  virtual int Initialize() = 0;
  virtual bool IsInitialized() const = 0;
  virtual int Compute() = 0;
  virtual bool IsComputed() const = 0;
  virtual double Condest(const Ifpack_CondestType CT = Ifpack_Cheap, const int MaxIters = 1550, const double Tol = 1e-9,
                         Epetra_RowMatrix* Matrix = 0) = 0;
  virtual double Condest() const = 0;
  virtual int ApplyInverse(const Epetra_MultiVector& X, Epetra_MultiVector& Y) const = 0;
  virtual const Epetra_RowMatrix& Matrix() const = 0;
  virtual int NumInitialize() const = 0;
  virtual int NumCompute() const = 0;
  virtual int NumApplyInverse() const = 0;
  virtual double InitializeTime() const = 0;
  virtual double ComputeTime() const = 0;
  virtual double ApplyInverseTime() const = 0;
  virtual double InitializeFlops() const = 0;
  virtual double ComputeFlops() const = 0;
  virtual double ApplyInverseFlops() const = 0;
  virtual std::ostream& Print(std::ostream& os) const = 0;
};
inline std::ostream& operator<<(std::ostream& os, const Ifpack_Preconditioner& obj) { return obj.Print(os); }
#endif
