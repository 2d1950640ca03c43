!> Exercises part 2 of the Fortran module layer (fortran/ntpoly_amd_modules_more.f90) the way a Fortran user of NTPoly
!> writes it: module, type and procedure names are the reference's.  Builds a banded SPD matrix from triplets and checks
!> Cholesky (L L^T = A), CG (A X = B), the eigendecomposition (V W V^T = A), the Pade exponential against the
!> Chebyshev-free Taylor-type exponential, a matrix polynomial (Horner vs Paterson-Stockmeyer), the load balancer
!> round trip and the dense inverse.  Prints one line per check and "ALL PASS" at the end.
PROGRAM SolverFamilies
  USE DataTypesModule, ONLY : NTREAL
  USE ProcessGridModule, ONLY : ConstructProcessGrid, DestructProcessGrid
  USE TripletListModule, ONLY : Triplet_r, TripletList_r, ConstructTripletList, AppendToTripletList, DestructTripletList
  USE PSMatrixModule, ONLY : Matrix_ps, ConstructEmptyMatrix, FillMatrixFromTripletList, FillMatrixIdentity, &
       & DestructMatrix, TransposeMatrix, CopyMatrix
  USE PSMatrixAlgebraModule, ONLY : MatrixMultiply, IncrementMatrix, MatrixNorm, ScaleMatrix
  USE SolverParametersModule, ONLY : SolverParameters_t, ConstructSolverParameters
  USE PermutationModule, ONLY : Permutation_t, ConstructRandomPermutation, DestructPermutation
  USE LinearSolversModule, ONLY : CGSolver, CholeskyDecomposition
  USE EigenSolversModule, ONLY : EigenDecomposition
  USE ExponentialSolversModule, ONLY : ComputeExponential, ComputeExponentialPade
  USE PolynomialSolversModule, ONLY : Polynomial_t, ConstructPolynomial, SetCoefficient, Compute, FactorizedCompute
  USE LoadBalancerModule, ONLY : PermuteMatrix, UndoPermuteMatrix
  USE DenseSolversModule, ONLY : DenseInvert
  IMPLICIT NONE
  INTEGER, PARAMETER :: n = 120, h = 4
  TYPE(Matrix_ps) :: A, L, LT, T, X, B, W, V, VT, E1, E2, P1, P2, Ainv, Ident
  TYPE(TripletList_r) :: tl
  TYPE(Triplet_r) :: trip
  TYPE(SolverParameters_t) :: sp, sp0
  TYPE(Polynomial_t) :: poly
  TYPE(Permutation_t) :: perm
  INTEGER :: i, j, nfail
  REAL(NTREAL) :: err

  CALL ConstructProcessGrid(0, 1, 1, 1)
  nfail = 0
  CALL ConstructTripletList(tl)
  DO j = 1, n
     DO i = MAX(1, j - h), MIN(n, j + h)
        trip%index_column = j
        trip%index_row = i
        IF (i .EQ. j) THEN
           trip%point_value = 3.0_NTREAL + 0.01_NTREAL * MOD(7 * j, 13)
        ELSE
           trip%point_value = -0.4_NTREAL / ABS(i - j)
        END IF
        CALL AppendToTripletList(tl, trip)
     END DO
  END DO
  CALL ConstructEmptyMatrix(A, n)
  CALL FillMatrixFromTripletList(A, tl)
  CALL DestructTripletList(tl)
  CALL ConstructEmptyMatrix(Ident, n)
  CALL FillMatrixIdentity(Ident)
  CALL ConstructSolverParameters(sp, threshold_in = 1e-12_NTREAL, converge_diff_in = 1e-10_NTREAL)
  CALL ConstructSolverParameters(sp0, threshold_in = 0.0_NTREAL, converge_diff_in = 1e-10_NTREAL)

  ! Cholesky: L L^T = A
  CALL CholeskyDecomposition(A, L, sp)
  CALL TransposeMatrix(L, LT)
  CALL MatrixMultiply(L, LT, T, threshold_in = 0.0_NTREAL)
  CALL IncrementMatrix(A, T, alpha_in = -1.0_NTREAL)
  CALL report("cholesky", MatrixNorm(T), 1e-10_NTREAL)

  ! CG: A X = I  and the dense inverse
  CALL CGSolver(A, X, Ident, sp0)
  CALL MatrixMultiply(A, X, T, threshold_in = 0.0_NTREAL)
  CALL IncrementMatrix(Ident, T, alpha_in = -1.0_NTREAL)
  CALL report("cg", MatrixNorm(T), 1e-7_NTREAL)
  CALL DenseInvert(A, Ainv, sp)
  CALL IncrementMatrix(X, Ainv, alpha_in = -1.0_NTREAL)
  CALL report("dense inverse", MatrixNorm(Ainv), 1e-7_NTREAL)

  ! eigendecomposition: V W V^T = A
  CALL EigenDecomposition(A, W, eigenvectors_in = V, solver_parameters_in = sp)
  CALL TransposeMatrix(V, VT)
  CALL MatrixMultiply(V, W, T, threshold_in = 0.0_NTREAL)
  CALL MatrixMultiply(T, VT, B, threshold_in = 0.0_NTREAL)
  CALL IncrementMatrix(A, B, alpha_in = -1.0_NTREAL)
  CALL report("eigendecomposition", MatrixNorm(B), 1e-9_NTREAL)

  ! exponential: Pade vs the default (Chebyshev) solver, on A / 4
  CALL CopyMatrix(A, T)
  CALL ScaleMatrix(T, 0.25_NTREAL)
  CALL ComputeExponential(T, E1, sp)
  CALL ComputeExponentialPade(T, E2, sp0)
  CALL IncrementMatrix(E1, E2, alpha_in = -1.0_NTREAL)
  CALL report("exponential", MatrixNorm(E2) / MatrixNorm(E1), 1e-6_NTREAL)

  ! polynomial 1 + 0.5 x - 0.25 x^2 + 0.125 x^3: Horner vs Paterson-Stockmeyer
  CALL ConstructPolynomial(poly, 4)
  CALL SetCoefficient(poly, 1, 1.0_NTREAL)
  CALL SetCoefficient(poly, 2, 0.5_NTREAL)
  CALL SetCoefficient(poly, 3, -0.25_NTREAL)
  CALL SetCoefficient(poly, 4, 0.125_NTREAL)
  CALL Compute(T, P1, poly, sp)
  CALL FactorizedCompute(T, P2, poly, sp)
  CALL IncrementMatrix(P1, P2, alpha_in = -1.0_NTREAL)
  CALL report("polynomial", MatrixNorm(P2), 1e-10_NTREAL)

  ! load balancer round trip
  CALL ConstructRandomPermutation(perm, n)
  CALL PermuteMatrix(A, T, perm)
  CALL UndoPermuteMatrix(T, B, perm)
  CALL IncrementMatrix(A, B, alpha_in = -1.0_NTREAL)
  CALL report("load balancer", MatrixNorm(B), 1e-14_NTREAL)
  CALL DestructPermutation(perm)

  IF (nfail .EQ. 0) WRITE(*, '(A)') "ALL PASS"
  CALL DestructProcessGrid
CONTAINS
  SUBROUTINE report(name, val, tol)
    CHARACTER(len=*), INTENT(IN) :: name
    REAL(NTREAL), INTENT(IN) :: val, tol
    IF (val .LE. tol) THEN
       WRITE(*, '(A,A,ES10.2)') "ok   ", name, val
    ELSE
       WRITE(*, '(A,A,ES10.2)') "FAIL ", name, val
       nfail = nfail + 1
    END IF
  END SUBROUTINE report
END PROGRAM SolverFamilies
