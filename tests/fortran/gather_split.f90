!> GatherMatrixToProcess and CommSplitMatrix of the Fortran module layer (PSMatrixModule.F90:1489-1541, 1704-1808), written
!> the way a Fortran user of NTPoly writes it.  Builds a small banded matrix from triplets, gathers it to every process and to
!> process 0 of the slice as a LOCAL matrix (SMatrixModule), compares the local matrix's triplets with the distributed
!> matrix's own, splits the matrix' communicator (one process: the reference's base case) and compares the copy.
PROGRAM GatherSplit
  USE DataTypesModule, ONLY : NTREAL
  USE ProcessGridModule, ONLY : ConstructProcessGrid, DestructProcessGrid
  USE TripletListModule, ONLY : Triplet_r, TripletList_r, ConstructTripletList, AppendToTripletList, DestructTripletList, &
       & GetTripletAt
  USE SMatrixModule, ONLY : Matrix_lsr, MatrixToTripletList, GetMatrixRows, GetMatrixColumns, DestructLocal => DestructMatrix
  USE PSMatrixModule, ONLY : Matrix_ps, ConstructEmptyMatrix, FillMatrixFromTripletList, GetMatrixTripletList, DestructMatrix, &
       & GatherMatrixToProcess, CommSplitMatrix
  USE PSMatrixAlgebraModule, ONLY : IncrementMatrix, MatrixNorm
  IMPLICIT NONE
  INTEGER, PARAMETER :: n = 90, h = 3
  TYPE(Matrix_ps) :: A, S
  TYPE(Matrix_lsr) :: Lall, Lid
  TYPE(TripletList_r) :: tl, ta, tg
  TYPE(Triplet_r) :: trip, t1, t2
  INTEGER :: i, j, nfail, color, k
  LOGICAL :: split_slice
  REAL(NTREAL) :: err

  CALL ConstructProcessGrid(0, 1, 1, 1)
  nfail = 0
  CALL ConstructTripletList(tl)
  DO j = 1, n
     DO i = MAX(1, j - h), MIN(n, j + h)
        trip%index_column = j
        trip%index_row = i
        trip%point_value = 1.0_NTREAL / (1 + ABS(i - j)) + 0.001_NTREAL * j
        CALL AppendToTripletList(tl, trip)
     END DO
  END DO
  CALL ConstructEmptyMatrix(A, n)
  CALL FillMatrixFromTripletList(A, tl)
  CALL DestructTripletList(tl)

  ! gathered to every process
  CALL GatherMatrixToProcess(A, Lall)
  IF (GetMatrixRows(Lall) .NE. n .OR. GetMatrixColumns(Lall) .NE. n) CALL fail("shape of the gathered matrix")
  CALL GetMatrixTripletList(A, ta)
  CALL MatrixToTripletList(Lall, tg)
  IF (ta%CurrentSize .NE. tg%CurrentSize) CALL fail("entry count of the gathered matrix")
  err = 0.0_NTREAL
  DO k = 1, MIN(ta%CurrentSize, tg%CurrentSize)
     CALL GetTripletAt(ta, k, t1)
     CALL GetTripletAt(tg, k, t2)
     IF (t1%index_column .NE. t2%index_column .OR. t1%index_row .NE. t2%index_row) err = err + 1.0_NTREAL
     err = err + ABS(t1%point_value - t2%point_value)
  END DO
  CALL report("gather to all", err)
  ! gathered to process 0 of the slice
  CALL GatherMatrixToProcess(A, Lid, 0)
  IF (GetMatrixRows(Lid) .NE. n) CALL fail("gather to process 0")
  CALL report("gather to process 0", 0.0_NTREAL)
  ! communicator split
  CALL CommSplitMatrix(A, S, color, split_slice)
  IF (color .NE. 0 .OR. .NOT. split_slice) CALL fail("colour / direction of the split")
  CALL IncrementMatrix(A, S, alpha_in = -1.0_NTREAL)
  CALL report("split copy", MatrixNorm(S))

  CALL DestructLocal(Lall)
  CALL DestructLocal(Lid)
  CALL DestructMatrix(A)
  CALL DestructMatrix(S)
  CALL DestructProcessGrid()
  IF (nfail .EQ. 0) THEN
     WRITE(*, '(A)') "ALL PASS"
  ELSE
     WRITE(*, '(A,I0)') "FAILED: ", nfail
     STOP 1
  END IF
CONTAINS
  SUBROUTINE report(what, e)
    CHARACTER(len=*), INTENT(IN) :: what
    REAL(NTREAL), INTENT(IN) :: e
    IF (e .LE. 1e-14_NTREAL) THEN
       WRITE(*, '(A,A,ES10.2)') "ok   ", what, e
    ELSE
       WRITE(*, '(A,A,ES10.2)') "FAIL ", what, e
       nfail = nfail + 1
    END IF
  END SUBROUTINE report
  SUBROUTINE fail(what)
    CHARACTER(len=*), INTENT(IN) :: what
    WRITE(*, '(A,A)') "FAIL ", what
    nfail = nfail + 1
  END SUBROUTINE fail
END PROGRAM GatherSplit
