// mock: the pure virtuals of Epetra_Operator (Trilinos packages/epetra/src/Epetra_Operator.h)
#ifndef MOCK_EPETRA_OPERATOR_H
#define MOCK_EPETRA_OPERATOR_H
#include "Epetra_Comm.h"
#include "Epetra_Map.h"
#include "Epetra_MultiVector.h"
class Epetra_Operator {
 public:
  virtual ~Epetra_Operator() {}
  virtual int SetUseTranspose(bool UseTranspose) = 0;
  virtual int Apply(const Epetra_MultiVector& X, Epetra_MultiVector& Y) const = 0;
  virtual int ApplyInverse(const Epetra_MultiVector& X, Epetra_MultiVector& Y) const = 0;
  virtual double NormInf() const = 0;
  virtual const char* Label() const = 0;
  virtual bool UseTranspose() const = 0;
  virtual bool HasNormInf() const = 0;
  virtual const Epetra_Comm& Comm() const = 0;
  virtual const Epetra_Map& OperatorDomainMap() const = 0;
  virtual const Epetra_Map& OperatorRangeMap() const = 0;
};
#endif
