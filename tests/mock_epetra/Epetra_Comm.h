// mock: the two members of Epetra_Comm the adapter reads
#ifndef MOCK_EPETRA_COMM_H
#define MOCK_EPETRA_COMM_H
class Epetra_Comm {
 public:
  virtual ~Epetra_Comm() {}
  virtual int MyPID() const = 0;
  virtual int NumProc() const = 0;
};
class Epetra_SerialComm : public Epetra_Comm {
 public:
  int MyPID() const { return 0; }
  int NumProc() const { return 1; }
};
#endif
