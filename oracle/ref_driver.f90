!! TEST INFRASTRUCTURE ONLY -- our own driver program, linked against the REAL
!! NTPoly reference library built by oracle/build_ref.py into oracle/_ref/.
!! It exercises the reference's public Fortran module API on matrices stored in
!! a trivial binary triplet format (see tests/golden/make_golden.py) so that
!! golden input/output vectors can be produced bit-exactly, without going
!! through MatrixMarket text.  Nothing here is copied from the reference.
!!
!! File format ("tri"): int32 rows, cols, nnz, is_complex; then
!!   int32 col(nnz) ; int32 row(nnz) ; f64 val(nnz)  [or 2*nnz doubles re,im]
!! indices are 1-based (NTPoly triplet convention: index_column, index_row).
!!
!! usage: ref_driver <command> args...   (see SELECT CASE below)
PROGRAM RefDriver
  USE DataTypesModule, ONLY : NTREAL, NTCOMPLEX
  USE TripletListModule, ONLY : TripletList_r, TripletList_c, &
       & ConstructTripletList, DestructTripletList
  USE SMatrixModule, ONLY : Matrix_lsr, Matrix_lsc, &
       & ConstructMatrixFromTripletList, MatrixToTripletList, &
       & DestructMatrix, TransposeMatrix
  USE SMatrixAlgebraModule, ONLY : MatrixMultiply, IncrementMatrix, DotMatrix
  USE ProcessGridModule, ONLY : ConstructProcessGrid, DestructProcessGrid, IsRoot
  USE PSMatrixModule, ONLY : Matrix_ps, ConstructEmptyMatrix, &
       & FillMatrixFromTripletList, GetMatrixTripletList, FillMatrixIdentity, &
       & CopyMatrix, GetMatrixSize
  USE PSMatrixAlgebraModule, ONLY : PMultiply => MatrixMultiply, &
       & PIncrement => IncrementMatrix, PDot => DotMatrix, MatrixTrace, &
       & MatrixNorm, ScaleMatrix
  USE PMatrixMemoryPoolModule, ONLY : MatrixMemoryPool_p
  USE EigenBoundsModule, ONLY : GershgorinBounds
  USE SolverParametersModule, ONLY : SolverParameters_t, &
       & ConstructSolverParameters
  USE PermutationModule, ONLY : Permutation_t, ConstructDefaultPermutation
  USE DensityMatrixSolversModule, ONLY : TRS2, TRS4, PM, HPCP, ScaleAndFold
  USE SignSolversModule, ONLY : SignFunction, PolarDecomposition
  USE InverseSolversModule, ONLY : Invert, PseudoInverse
  USE SquareRootSolversModule, ONLY : InverseSquareRoot, SquareRoot
  USE LoggingModule, ONLY : ActivateLogger, DeactivateLogger
  USE PolynomialSolversModule, ONLY : Polynomial_t, ConstructPolynomial, &
       & SetCoefficient, HornerCompute => Compute, &
       & PatersonStockmeyerCompute => FactorizedCompute
  USE ChebyshevSolversModule, ONLY : ChebyshevPolynomial_t, &
       & ChebConstruct => ConstructPolynomial, ChebSet => SetCoefficient, &
       & ChebCompute => Compute, ChebFactorized => FactorizedCompute
  USE ExponentialSolversModule, ONLY : ComputeExponential, ComputeLogarithm
  USE TrigonometrySolversModule, ONLY : Sine, Cosine
  USE RootSolversModule, ONLY : ComputeRoot, ComputeInverseRoot
  USE EigenBoundsModule, ONLY : PowerBounds
  USE LinearSolversModule, ONLY : CGSolver, CholeskyDecomposition
  USE AnalysisModule, ONLY : PivotedCholeskyDecomposition, ReduceDimension
  USE ExponentialSolversModule, ONLY : ComputeExponentialPade, &
       & ComputeDenseExponential, ComputeDenseLogarithm
  USE GeometryOptimizationModule, ONLY : PurificationExtrapolate, &
       & LowdinExtrapolate
  USE MatrixConversionModule, ONLY : SnapMatrixToSparsityPattern
  USE EigenSolversModule, ONLY : EigenDecomposition, EstimateGap
  USE SingularValueSolversModule, ONLY : SingularValueDecomposition
  USE FermiOperatorModule, ONLY : ComputeDenseFOE, WOM_GC, WOM_C
  USE DensityMatrixSolversModule, ONLY : DenseDensity
  USE SquareRootSolversModule, ONLY : DenseSquareRoot, DenseInverseSquareRoot
  USE TrigonometrySolversModule, ONLY : DenseSine, DenseCosine
  USE InverseSolversModule, ONLY : DenseInvert
  USE SignSolversModule, ONLY : DenseSignFunction
  USE HermiteSolversModule, ONLY : HermitePolynomial_t, &
       & HermConstruct => ConstructPolynomial, HermSet => SetCoefficient, &
       & HermCompute => Compute
  IMPLICIT NONE
  INCLUDE "mpif.h"
  CHARACTER(len=32) :: cmd
  INTEGER :: provided, ierr

  CALL MPI_Init_thread(MPI_THREAD_SERIALIZED, provided, ierr)
  CALL GET_COMMAND_ARGUMENT(1, cmd)
  SELECT CASE(TRIM(cmd))
  CASE("lgemm")
     CALL cmd_lgemm()
  CASE("lincr")
     CALL cmd_lincr()
  CASE("pgemm")
     CALL cmd_pgemm()
  CASE("pincr")
     CALL cmd_pincr()
  CASE("pscalars")
     CALL cmd_pscalars()
  CASE("solve")
     CALL cmd_solve()
  CASE("poly")
     CALL cmd_poly()
  CASE("func")
     CALL cmd_func()
  CASE("extra")
     CALL cmd_extra()
  CASE DEFAULT
     WRITE(*,*) "unknown command ", cmd
  END SELECT
  CALL MPI_Finalize(ierr)
CONTAINS
  FUNCTION sarg(i) RESULT(s)
    INTEGER, INTENT(IN) :: i
    CHARACTER(len=256) :: s
    CALL GET_COMMAND_ARGUMENT(i, s)
  END FUNCTION sarg
  FUNCTION rarg(i) RESULT(r)
    INTEGER, INTENT(IN) :: i
    REAL(NTREAL) :: r
    CHARACTER(len=256) :: s
    CALL GET_COMMAND_ARGUMENT(i, s)
    READ(s, *) r
  END FUNCTION rarg
  FUNCTION iarg(i) RESULT(r)
    INTEGER, INTENT(IN) :: i
    INTEGER :: r
    CHARACTER(len=256) :: s
    CALL GET_COMMAND_ARGUMENT(i, s)
    READ(s, *) r
  END FUNCTION iarg

  SUBROUTINE read_header(fname, rows, cols, nnz, is_complex)
    CHARACTER(len=*), INTENT(IN) :: fname
    INTEGER, INTENT(OUT) :: rows, cols, nnz, is_complex
    INTEGER :: u
    OPEN(NEWUNIT=u, FILE=TRIM(fname), ACCESS="STREAM", FORM="UNFORMATTED", &
         & STATUS="OLD")
    READ(u) rows, cols, nnz, is_complex
    CLOSE(u)
  END SUBROUTINE read_header

  SUBROUTINE read_tri_r(fname, tl, rows, cols)
    CHARACTER(len=*), INTENT(IN) :: fname
    TYPE(TripletList_r), INTENT(INOUT) :: tl
    INTEGER, INTENT(OUT) :: rows, cols
    INTEGER :: u, nnz, is_complex, II
    INTEGER, ALLOCATABLE :: ci(:), ri(:)
    REAL(NTREAL), ALLOCATABLE :: v(:)
    OPEN(NEWUNIT=u, FILE=TRIM(fname), ACCESS="STREAM", FORM="UNFORMATTED", &
         & STATUS="OLD")
    READ(u) rows, cols, nnz, is_complex
    ALLOCATE(ci(nnz), ri(nnz), v(nnz))
    IF (nnz .GT. 0) READ(u) ci, ri, v
    CLOSE(u)
    CALL ConstructTripletList(tl, nnz)
    DO II = 1, nnz
       tl%DATA(II)%index_column = ci(II)
       tl%DATA(II)%index_row = ri(II)
       tl%DATA(II)%point_value = v(II)
    END DO
  END SUBROUTINE read_tri_r

  SUBROUTINE read_tri_c(fname, tl, rows, cols)
    CHARACTER(len=*), INTENT(IN) :: fname
    TYPE(TripletList_c), INTENT(INOUT) :: tl
    INTEGER, INTENT(OUT) :: rows, cols
    INTEGER :: u, nnz, is_complex, II
    INTEGER, ALLOCATABLE :: ci(:), ri(:)
    REAL(NTREAL), ALLOCATABLE :: v(:)
    OPEN(NEWUNIT=u, FILE=TRIM(fname), ACCESS="STREAM", FORM="UNFORMATTED", &
         & STATUS="OLD")
    READ(u) rows, cols, nnz, is_complex
    ALLOCATE(ci(nnz), ri(nnz), v(2*nnz))
    IF (nnz .GT. 0) READ(u) ci, ri, v
    CLOSE(u)
    CALL ConstructTripletList(tl, nnz)
    DO II = 1, nnz
       tl%DATA(II)%index_column = ci(II)
       tl%DATA(II)%index_row = ri(II)
       tl%DATA(II)%point_value = CMPLX(v(2*II-1), v(2*II), KIND=NTCOMPLEX)
    END DO
  END SUBROUTINE read_tri_c

  SUBROUTINE write_tri_r(fname, tl, rows, cols)
    CHARACTER(len=*), INTENT(IN) :: fname
    TYPE(TripletList_r), INTENT(IN) :: tl
    INTEGER, INTENT(IN) :: rows, cols
    INTEGER :: u, nnz, II
    INTEGER, ALLOCATABLE :: ci(:), ri(:)
    REAL(NTREAL), ALLOCATABLE :: v(:)
    nnz = tl%CurrentSize
    ALLOCATE(ci(nnz), ri(nnz), v(nnz))
    DO II = 1, nnz
       ci(II) = tl%DATA(II)%index_column
       ri(II) = tl%DATA(II)%index_row
       v(II) = tl%DATA(II)%point_value
    END DO
    OPEN(NEWUNIT=u, FILE=TRIM(fname), ACCESS="STREAM", FORM="UNFORMATTED", &
         & STATUS="REPLACE")
    WRITE(u) rows, cols, nnz, 0
    IF (nnz .GT. 0) WRITE(u) ci, ri, v
    CLOSE(u)
  END SUBROUTINE write_tri_r

  SUBROUTINE write_tri_c(fname, tl, rows, cols)
    CHARACTER(len=*), INTENT(IN) :: fname
    TYPE(TripletList_c), INTENT(IN) :: tl
    INTEGER, INTENT(IN) :: rows, cols
    INTEGER :: u, nnz, II
    INTEGER, ALLOCATABLE :: ci(:), ri(:)
    REAL(NTREAL), ALLOCATABLE :: v(:)
    nnz = tl%CurrentSize
    ALLOCATE(ci(nnz), ri(nnz), v(2*nnz))
    DO II = 1, nnz
       ci(II) = tl%DATA(II)%index_column
       ri(II) = tl%DATA(II)%index_row
       v(2*II-1) = REAL(tl%DATA(II)%point_value, KIND=NTREAL)
       v(2*II) = AIMAG(tl%DATA(II)%point_value)
    END DO
    OPEN(NEWUNIT=u, FILE=TRIM(fname), ACCESS="STREAM", FORM="UNFORMATTED", &
         & STATUS="REPLACE")
    WRITE(u) rows, cols, nnz, 1
    IF (nnz .GT. 0) WRITE(u) ci, ri, v
    CLOSE(u)
  END SUBROUTINE write_tri_c

  !! Load a distributed matrix (only the root contributes triplets).
  SUBROUTINE load_ps(fname, mat)
    CHARACTER(len=*), INTENT(IN) :: fname
    TYPE(Matrix_ps), INTENT(INOUT) :: mat
    TYPE(TripletList_r) :: tr
    TYPE(TripletList_c) :: tc
    INTEGER :: rows, cols, nnz, is_complex
    CALL read_header(fname, rows, cols, nnz, is_complex)
    IF (is_complex .EQ. 1) THEN
       CALL ConstructEmptyMatrix(mat, rows, is_complex_in=.TRUE.)
       IF (IsRoot()) THEN
          CALL read_tri_c(fname, tc, rows, cols)
       ELSE
          CALL ConstructTripletList(tc)
       END IF
       CALL FillMatrixFromTripletList(mat, tc)
    ELSE
       CALL ConstructEmptyMatrix(mat, rows)
       IF (IsRoot()) THEN
          CALL read_tri_r(fname, tr, rows, cols)
       ELSE
          CALL ConstructTripletList(tr)
       END IF
       CALL FillMatrixFromTripletList(mat, tr)
    END IF
  END SUBROUTINE load_ps

  !! Store a distributed matrix; with >1 rank each rank writes fname.<rank>.
  SUBROUTINE store_ps(fname, mat)
    CHARACTER(len=*), INTENT(IN) :: fname
    TYPE(Matrix_ps), INTENT(IN) :: mat
    TYPE(TripletList_r) :: tr
    TYPE(TripletList_c) :: tc
    CHARACTER(len=300) :: fn
    INTEGER :: rank, nranks, ierr2
    CALL MPI_Comm_rank(MPI_COMM_WORLD, rank, ierr2)
    CALL MPI_Comm_size(MPI_COMM_WORLD, nranks, ierr2)
    IF (nranks .GT. 1) THEN
       WRITE(fn, '(A,A,I0)') TRIM(fname), ".", rank
    ELSE
       fn = fname
    END IF
    IF (mat%is_complex) THEN
       CALL GetMatrixTripletList(mat, tc)
       CALL write_tri_c(fn, tc, mat%actual_matrix_dimension, &
            & mat%actual_matrix_dimension)
    ELSE
       CALL GetMatrixTripletList(mat, tr)
       CALL write_tri_r(fn, tr, mat%actual_matrix_dimension, &
            & mat%actual_matrix_dimension)
    END IF
  END SUBROUTINE store_ps

  SUBROUTINE make_grid(first_arg)
    INTEGER, INTENT(IN) :: first_arg
    CALL ConstructProcessGrid(MPI_COMM_WORLD, iarg(first_arg), &
         & iarg(first_arg + 1), iarg(first_arg + 2))
  END SUBROUTINE make_grid

  !! lgemm A B Cin|none tA tB alpha beta thr out     (local, SMatrix level)
  SUBROUTINE cmd_lgemm()
    TYPE(TripletList_r) :: tr
    TYPE(TripletList_c) :: tc
    TYPE(Matrix_lsr) :: Ar, Br, Cr
    TYPE(Matrix_lsc) :: Ac, Bc, Cc
    INTEGER :: rows, cols, nnz, is_complex, crows, ccols
    LOGICAL :: tA, tB
    REAL(NTREAL) :: alpha, beta, thr
    tA = iarg(5) .NE. 0
    tB = iarg(6) .NE. 0
    alpha = rarg(7); beta = rarg(8); thr = rarg(9)
    CALL read_header(sarg(2), rows, cols, nnz, is_complex)
    IF (is_complex .EQ. 1) THEN
       CALL read_tri_c(sarg(2), tc, rows, cols)
       CALL ConstructMatrixFromTripletList(Ac, tc, rows, cols)
       crows = MERGE(cols, rows, tA)
       CALL read_tri_c(sarg(3), tc, rows, cols)
       CALL ConstructMatrixFromTripletList(Bc, tc, rows, cols)
       ccols = MERGE(rows, cols, tB)
       IF (TRIM(sarg(4)) .NE. "none") THEN
          CALL read_tri_c(sarg(4), tc, rows, cols)
          CALL ConstructMatrixFromTripletList(Cc, tc, rows, cols)
          CALL MatrixMultiply(Ac, Bc, Cc, IsATransposed_in=tA, &
               & IsBTransposed_in=tB, alpha_in=alpha, beta_in=beta, &
               & threshold_in=thr)
       ELSE
          CALL MatrixMultiply(Ac, Bc, Cc, IsATransposed_in=tA, &
               & IsBTransposed_in=tB, alpha_in=alpha, threshold_in=thr)
       END IF
       CALL MatrixToTripletList(Cc, tc)
       CALL write_tri_c(sarg(10), tc, crows, ccols)
    ELSE
       CALL read_tri_r(sarg(2), tr, rows, cols)
       CALL ConstructMatrixFromTripletList(Ar, tr, rows, cols)
       crows = MERGE(cols, rows, tA)
       CALL read_tri_r(sarg(3), tr, rows, cols)
       CALL ConstructMatrixFromTripletList(Br, tr, rows, cols)
       ccols = MERGE(rows, cols, tB)
       IF (TRIM(sarg(4)) .NE. "none") THEN
          CALL read_tri_r(sarg(4), tr, rows, cols)
          CALL ConstructMatrixFromTripletList(Cr, tr, rows, cols)
          CALL MatrixMultiply(Ar, Br, Cr, IsATransposed_in=tA, &
               & IsBTransposed_in=tB, alpha_in=alpha, beta_in=beta, &
               & threshold_in=thr)
       ELSE
          CALL MatrixMultiply(Ar, Br, Cr, IsATransposed_in=tA, &
               & IsBTransposed_in=tB, alpha_in=alpha, threshold_in=thr)
       END IF
       CALL MatrixToTripletList(Cr, tr)
       CALL write_tri_r(sarg(10), tr, crows, ccols)
    END IF
  END SUBROUTINE cmd_lgemm

  !! lincr A B alpha thr out    (local B <- alpha*A + B)
  SUBROUTINE cmd_lincr()
    TYPE(TripletList_r) :: tr
    TYPE(TripletList_c) :: tc
    TYPE(Matrix_lsr) :: Ar, Br
    TYPE(Matrix_lsc) :: Ac, Bc
    INTEGER :: rows, cols, nnz, is_complex
    CALL read_header(sarg(2), rows, cols, nnz, is_complex)
    IF (is_complex .EQ. 1) THEN
       CALL read_tri_c(sarg(2), tc, rows, cols)
       CALL ConstructMatrixFromTripletList(Ac, tc, rows, cols)
       CALL read_tri_c(sarg(3), tc, rows, cols)
       CALL ConstructMatrixFromTripletList(Bc, tc, rows, cols)
       CALL IncrementMatrix(Ac, Bc, alpha_in=rarg(4), threshold_in=rarg(5))
       CALL MatrixToTripletList(Bc, tc)
       CALL write_tri_c(sarg(6), tc, rows, cols)
    ELSE
       CALL read_tri_r(sarg(2), tr, rows, cols)
       CALL ConstructMatrixFromTripletList(Ar, tr, rows, cols)
       CALL read_tri_r(sarg(3), tr, rows, cols)
       CALL ConstructMatrixFromTripletList(Br, tr, rows, cols)
       CALL IncrementMatrix(Ar, Br, alpha_in=rarg(4), threshold_in=rarg(5))
       CALL MatrixToTripletList(Br, tr)
       CALL write_tri_r(sarg(6), tr, rows, cols)
    END IF
  END SUBROUTINE cmd_lincr

  !! pgemm pr pc ps A B Cin|none alpha beta thr out
  SUBROUTINE cmd_pgemm()
    TYPE(Matrix_ps) :: A, B, C
    TYPE(MatrixMemoryPool_p) :: pool
    CALL make_grid(2)
    CALL load_ps(sarg(5), A)
    CALL load_ps(sarg(6), B)
    IF (TRIM(sarg(7)) .NE. "none") THEN
       CALL load_ps(sarg(7), C)
       CALL PMultiply(A, B, C, alpha_in=rarg(8), beta_in=rarg(9), &
            & threshold_in=rarg(10), memory_pool_in=pool)
    ELSE
       CALL PMultiply(A, B, C, alpha_in=rarg(8), threshold_in=rarg(10), &
            & memory_pool_in=pool)
    END IF
    CALL store_ps(sarg(11), C)
    CALL DestructProcessGrid
  END SUBROUTINE cmd_pgemm

  !! pincr pr pc ps A B alpha thr out
  SUBROUTINE cmd_pincr()
    TYPE(Matrix_ps) :: A, B
    CALL make_grid(2)
    CALL load_ps(sarg(5), A)
    CALL load_ps(sarg(6), B)
    CALL PIncrement(A, B, alpha_in=rarg(7), threshold_in=rarg(8))
    CALL store_ps(sarg(9), B)
    CALL DestructProcessGrid
  END SUBROUTINE cmd_pincr

  !! pscalars pr pc ps A B out.txt : trace(A) norm(A) dot(A,B) gersh(A) nnz(A)
  SUBROUTINE cmd_pscalars()
    TYPE(Matrix_ps) :: A, B
    REAL(NTREAL) :: tr, nrm, dt, emin, emax
    COMPLEX(NTCOMPLEX) :: dtc
    INTEGER :: u
    CALL make_grid(2)
    CALL load_ps(sarg(5), A)
    CALL load_ps(sarg(6), B)
    CALL MatrixTrace(A, tr)
    nrm = MatrixNorm(A)
    dtc = 0
    IF (A%is_complex) THEN
       CALL PDot(A, B, dtc)
       dt = REAL(dtc, KIND=NTREAL)
    ELSE
       CALL PDot(A, B, dt)
    END IF
    CALL GershgorinBounds(A, emin, emax)
    IF (IsRoot()) THEN
       OPEN(NEWUNIT=u, FILE=TRIM(sarg(7)), STATUS="REPLACE")
       WRITE(u, '(A,ES26.17E3)') "trace ", tr
       WRITE(u, '(A,ES26.17E3)') "norm ", nrm
       WRITE(u, '(A,ES26.17E3)') "dot_real ", dt
       WRITE(u, '(A,ES26.17E3)') "dot_imag ", AIMAG(dtc)
       WRITE(u, '(A,ES26.17E3)') "gersh_min ", emin
       WRITE(u, '(A,ES26.17E3)') "gersh_max ", emax
       WRITE(u, '(A,I0)') "nnz ", GetMatrixSize(A)
       CLOSE(u)
    END IF
    CALL DestructProcessGrid
  END SUBROUTINE cmd_pscalars

  !! poly pr pc ps <kind> A thr out ncoef c_1 ... c_ncoef
  !!   kind in {horner, ps, cheby, chebyfact, hermite}; coefficient i multiplies x^(i-1) / T_(i-1) / H_(i-1)
  SUBROUTINE cmd_poly()
    TYPE(Matrix_ps) :: A, K
    TYPE(SolverParameters_t) :: sp
    TYPE(Polynomial_t) :: p1
    TYPE(ChebyshevPolynomial_t) :: p2
    TYPE(HermitePolynomial_t) :: p3
    CHARACTER(len=32) :: kind
    INTEGER :: n, II
    CALL make_grid(2)
    kind = sarg(5)
    CALL load_ps(sarg(6), A)
    CALL ConstructSolverParameters(sp, threshold_in=rarg(7))
    n = iarg(9)
    SELECT CASE(TRIM(kind))
    CASE("horner", "ps")
       CALL ConstructPolynomial(p1, n)
       DO II = 1, n
          CALL SetCoefficient(p1, II, rarg(9 + II))
       END DO
       IF (TRIM(kind) .EQ. "horner") THEN
          CALL HornerCompute(A, K, p1, sp)
       ELSE
          CALL PatersonStockmeyerCompute(A, K, p1, sp)
       END IF
    CASE("cheby", "chebyfact")
       CALL ChebConstruct(p2, n)
       DO II = 1, n
          CALL ChebSet(p2, II, rarg(9 + II))
       END DO
       IF (TRIM(kind) .EQ. "cheby") THEN
          CALL ChebCompute(A, K, p2, sp)
       ELSE
          CALL ChebFactorized(A, K, p2, sp)
       END IF
    CASE("hermite")
       CALL HermConstruct(p3, n)
       DO II = 1, n
          CALL HermSet(p3, II, rarg(9 + II))
       END DO
       CALL HermCompute(A, K, p3, sp)
    END SELECT
    CALL store_ps(sarg(8), K)
    CALL DestructProcessGrid
  END SUBROUTINE cmd_poly

  !! func pr pc ps <kind> A thr conv out scal.txt [root]
  !!   kind in {exp, log, sin, cos, root, invroot, power}
  SUBROUTINE cmd_func()
    TYPE(Matrix_ps) :: A, K
    TYPE(SolverParameters_t) :: sp
    CHARACTER(len=32) :: kind
    REAL(NTREAL) :: bound
    INTEGER :: u
    CALL make_grid(2)
    kind = sarg(5)
    CALL load_ps(sarg(6), A)
    IF (IsRoot()) CALL ActivateLogger(start_document_in=.TRUE., &
         & file_name_in=TRIM(sarg(9))//".log")
    CALL ConstructSolverParameters(sp, threshold_in=rarg(7), converge_diff_in=rarg(8), &
         & be_verbose_in=.TRUE.)
    bound = 0
    SELECT CASE(TRIM(kind))
    CASE("exp")
       CALL ComputeExponential(A, K, sp)
    CASE("log")
       CALL ComputeLogarithm(A, K, sp)
    CASE("sin")
       CALL Sine(A, K, sp)
    CASE("cos")
       CALL Cosine(A, K, sp)
    CASE("root")
       CALL ComputeRoot(A, K, iarg(11), sp)
    CASE("invroot")
       CALL ComputeInverseRoot(A, K, iarg(11), sp)
    CASE("power")
       CALL PowerBounds(A, bound, sp)
       CALL CopyMatrix(A, K)
    END SELECT
    IF (IsRoot()) CALL DeactivateLogger
    CALL store_ps(sarg(9), K)
    IF (IsRoot()) THEN
       OPEN(NEWUNIT=u, FILE=TRIM(sarg(10)), STATUS="REPLACE")
       WRITE(u, '(A,ES26.17E3)') "bound ", bound
       CLOSE(u)
    END IF
    CALL DestructProcessGrid
  END SUBROUTINE cmd_func

  !! extra pr pc ps <kind> A B C thr conv out scal.txt p1 p2    (B, C may be "none"; second output: out.2)
  !!   kind in {cg, pade, purify, lowdin, snap, eig, svd, gap, foe, density, womgc, womc, chol, pchol, reduce,
  !!            dsqrt, disqrt, dexp, dlog, dsin, dcos, dinv, dsign}
  SUBROUTINE cmd_extra()
    TYPE(Matrix_ps) :: A, B, C, K, K2, K3
    TYPE(SolverParameters_t) :: sp
    CHARACTER(len=32) :: kind
    REAL(NTREAL) :: s1, s2, p1, p2
    INTEGER :: u
    LOGICAL :: have2
    CALL make_grid(2)
    kind = sarg(5)
    CALL load_ps(sarg(6), A)
    IF (TRIM(sarg(7)) .NE. "none") CALL load_ps(sarg(7), B)
    IF (TRIM(sarg(8)) .NE. "none") CALL load_ps(sarg(8), C)
    p1 = rarg(13)
    p2 = rarg(14)
    !! (EstimateGap writes a key without a value, which the flang runtime rejects: it runs with the logger off)
    IF (IsRoot() .AND. TRIM(kind) .NE. "gap") CALL ActivateLogger(start_document_in=.TRUE., &
         & file_name_in=TRIM(sarg(11))//".log")
    CALL ConstructSolverParameters(sp, threshold_in=rarg(9), converge_diff_in=rarg(10), &
         & be_verbose_in=(TRIM(kind) .NE. "gap"))
    s1 = 0; s2 = 0
    have2 = .FALSE.
    SELECT CASE(TRIM(kind))
    CASE("cg")
       CALL CGSolver(A, K, B, sp)
    CASE("pade")
       CALL ComputeExponentialPade(A, K, sp)
    CASE("purify")
       CALL PurificationExtrapolate(A, B, p1, K, sp)
    CASE("lowdin")
       CALL LowdinExtrapolate(A, B, C, K, sp)
    CASE("snap")
       CALL CopyMatrix(A, K)
       CALL SnapMatrixToSparsityPattern(K, B)
    CASE("eig")
       CALL EigenDecomposition(A, K, eigenvectors_in=K2, nvals_in=INT(p1), &
            & solver_parameters_in=sp)
       have2 = .TRUE.
    CASE("svd")
       CALL SingularValueDecomposition(A, K2, K3, K, sp)
       have2 = .TRUE.
    CASE("gap")
       CALL EstimateGap(A, B, p1, s1, sp)
       CALL CopyMatrix(A, K)
    CASE("foe")
       CALL ComputeDenseFOE(A, B, p1, K, inv_temp_in=p2, energy_value_out=s1, &
            & chemical_potential_out=s2, solver_parameters_in=sp)
    CASE("density")
       CALL DenseDensity(A, B, p1, K, energy_value_out=s1, &
            & chemical_potential_out=s2, solver_parameters_in=sp)
    CASE("womgc")
       CALL WOM_GC(A, B, K, p1, p2, energy_value_out=s1, solver_parameters_in=sp)
    CASE("womc")
       CALL WOM_C(A, B, K, p1, p2, energy_value_out=s1, solver_parameters_in=sp)
    CASE("chol")
       CALL CholeskyDecomposition(A, K, sp)
    CASE("pchol")
       CALL PivotedCholeskyDecomposition(A, K, INT(p1), sp)
    CASE("reduce")
       CALL ReduceDimension(A, INT(p1), K, sp)
    CASE("dsqrt")
       CALL DenseSquareRoot(A, K, sp)
    CASE("disqrt")
       CALL DenseInverseSquareRoot(A, K, sp)
    CASE("dexp")
       CALL ComputeDenseExponential(A, K, sp)
    CASE("dlog")
       CALL ComputeDenseLogarithm(A, K, sp)
    CASE("dsin")
       CALL DenseSine(A, K, sp)
    CASE("dcos")
       CALL DenseCosine(A, K, sp)
    CASE("dinv")
       CALL DenseInvert(A, K, sp)
    CASE("dsign")
       CALL DenseSignFunction(A, K, sp)
    END SELECT
    IF (IsRoot() .AND. TRIM(kind) .NE. "gap") CALL DeactivateLogger
    CALL store_ps(sarg(11), K)
    IF (have2) CALL store_ps(TRIM(sarg(11))//".2", K2)
    IF (IsRoot()) THEN
       OPEN(NEWUNIT=u, FILE=TRIM(sarg(12)), STATUS="REPLACE")
       WRITE(u, '(A,ES26.17E3)') "s1 ", s1
       WRITE(u, '(A,ES26.17E3)') "s2 ", s2
       WRITE(u, '(A,I0)') "nnz ", GetMatrixSize(K)
       CLOSE(u)
    END IF
    CALL DestructProcessGrid
  END SUBROUTINE cmd_extra

  FUNCTION env_real(name) RESULT(v)
    CHARACTER(len=*), INTENT(IN) :: name
    REAL(NTREAL) :: v
    CHARACTER(len=64) :: buf
    INTEGER :: stat
    CALL GET_ENVIRONMENT_VARIABLE(name, buf, STATUS=stat)
    v = 0
    IF (stat .EQ. 0) READ(buf, *) v
  END FUNCTION env_real

  !! solve pr pc ps <solver> H ISQ|identity|none trace thr conv maxit monitor out log scal.txt
  !!   solver in {trs2, trs4, pm, hpcp, scalefold, sign, polar, invert, pinv, isq, sqrt}
  SUBROUTINE cmd_solve()
    TYPE(Matrix_ps) :: H, ISQ, K
    TYPE(SolverParameters_t) :: sp
    REAL(NTREAL) :: energy, mu
    CHARACTER(len=32) :: solver
    INTEGER :: u
    INTEGER(KIND=8) :: knnz
    TYPE(Permutation_t) :: perm
    CHARACTER(len=512) :: permfile
    INTEGER :: pstat, pn, pi
    CALL make_grid(2)
    solver = sarg(5)
    CALL load_ps(sarg(6), H)
    IF (TRIM(sarg(7)) .EQ. "identity") THEN
       CALL ConstructEmptyMatrix(ISQ, H)
       CALL FillMatrixIdentity(ISQ)
    ELSE IF (TRIM(sarg(7)) .NE. "none") THEN
       CALL load_ps(sarg(7), ISQ)
    END IF
    IF (IsRoot()) CALL ActivateLogger(start_document_in=.TRUE., &
         & file_name_in=TRIM(sarg(14)))
    !! REF_PERM=<file>: a load-balancing permutation given explicitly (one 1-based index_lookup value per line), so that
    !! the engine can be driven with the very same permutation (the reference's own comes from the Fortran RNG)
    CALL GET_ENVIRONMENT_VARIABLE("REF_PERM", permfile, STATUS=pstat)
    IF (pstat .EQ. 0 .AND. LEN_TRIM(permfile) .GT. 0) THEN
       pn = H%actual_matrix_dimension
       ALLOCATE(perm%index_lookup(H%logical_matrix_dimension))
       ALLOCATE(perm%reverse_index_lookup(H%logical_matrix_dimension))
       DO pi = 1, H%logical_matrix_dimension   ! (rows of the padding stay where they are)
          perm%index_lookup(pi) = pi
       END DO
       OPEN(NEWUNIT=u, FILE=TRIM(permfile), STATUS="OLD")
       DO pi = 1, pn
          READ(u, *) perm%index_lookup(pi)
       END DO
       CLOSE(u)
       DO pi = 1, H%logical_matrix_dimension
          perm%reverse_index_lookup(perm%index_lookup(pi)) = pi
       END DO
       CALL ConstructSolverParameters(sp, converge_diff_in=rarg(10), &
            & threshold_in=rarg(9), max_iterations_in=iarg(11), &
            & be_verbose_in=.TRUE., monitor_convergence_in=(iarg(12) .NE. 0), &
            & BalancePermutation_in=perm)
    ELSE
       CALL ConstructSolverParameters(sp, converge_diff_in=rarg(10), &
            & threshold_in=rarg(9), max_iterations_in=iarg(11), &
            & be_verbose_in=.TRUE., monitor_convergence_in=(iarg(12) .NE. 0))
    END IF
    energy = 0; mu = 0
    SELECT CASE(TRIM(solver))
    CASE("trs2")
       CALL TRS2(H, ISQ, rarg(8), K, energy_value_out=energy, &
            & chemical_potential_out=mu, solver_parameters_in=sp)
    CASE("trs4")
       CALL TRS4(H, ISQ, rarg(8), K, energy_value_out=energy, &
            & chemical_potential_out=mu, solver_parameters_in=sp)
    CASE("pm")
       CALL PM(H, ISQ, rarg(8), K, energy_value_out=energy, &
            & chemical_potential_out=mu, solver_parameters_in=sp)
    CASE("hpcp")
       CALL HPCP(H, ISQ, rarg(8), K, energy_value_out=energy, &
            & chemical_potential_out=mu, solver_parameters_in=sp)
    CASE("scalefold")   ! homo / lumo estimates come through the environment (REF_HOMO, REF_LUMO)
       CALL ScaleAndFold(H, ISQ, rarg(8), K, env_real("REF_HOMO"), env_real("REF_LUMO"), &
            & energy_value_out=energy, solver_parameters_in=sp)
    CASE("pinv")
       CALL PseudoInverse(H, K, sp)
    CASE("polar")
       CALL PolarDecomposition(H, K, solver_parameters_in=sp)
    CASE("sign")
       CALL SignFunction(H, K, sp)
    CASE("invert")
       CALL Invert(H, K, sp)
    CASE("isq")
       CALL InverseSquareRoot(H, K, sp)
    CASE("sqrt")
       CALL SquareRoot(H, K, sp)
    END SELECT
    IF (IsRoot()) CALL DeactivateLogger
    CALL store_ps(sarg(13), K)
    knnz = GetMatrixSize(K)   ! collective: every rank calls it
    IF (IsRoot()) THEN
       OPEN(NEWUNIT=u, FILE=TRIM(sarg(15)), STATUS="REPLACE")
       WRITE(u, '(A,ES26.17E3)') "energy ", energy
       WRITE(u, '(A,ES26.17E3)') "mu ", mu
       WRITE(u, '(A,I0)') "nnz ", knnz
       CLOSE(u)
    END IF
    CALL DestructProcessGrid
  END SUBROUTINE cmd_solve
END PROGRAM RefDriver
