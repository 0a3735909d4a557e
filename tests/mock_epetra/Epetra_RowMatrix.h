#ifndef MOCK_EPETRA_ROWMATRIX_H
#define MOCK_EPETRA_ROWMATRIX_H
#include "Epetra_Operator.h"
class Epetra_RowMatrix : public virtual Epetra_Operator {
 public:
  virtual ~Epetra_RowMatrix() {}
  virtual int NumMyRows() const = 0;
  virtual int NumMyNonzeros() const = 0;
  virtual const Epetra_Map& RowMatrixRowMap() const = 0;
  virtual const Epetra_Map& RowMatrixColMap() const = 0;
};
#endif
