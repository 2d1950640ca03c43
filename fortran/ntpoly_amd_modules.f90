!> Fortran module layer of the MI355X engine: the reference's module names, derived types and generic
!! procedure names (Source/Fortran/*Module.F90) re-expressed as thin ISO_C_BINDING wrappers over the C ABI of
!! libntpoly_amd.so (include/*.h).  A program written against NTPoly's Fortran API -- e.g. the reference's own
!! Examples/PremadeMatrix/main.f90 -- compiles unchanged against these modules and runs on the GPU engine.
!! Covered: the modules on and next to the hot path (data types, process grid, distributed matrix + algebra,
!! triplet lists, permutations, solver parameters, logging, density / sign / inverse / square-root solvers,
!! eigenvalue bounds).  Matrices are opaque handles here; only the members user code commonly reads
!! (actual_matrix_dimension, logical_matrix_dimension, is_complex) are mirrored.
MODULE DataTypesModule
  USE, INTRINSIC :: ISO_C_BINDING, ONLY : c_double, c_long, c_int
  IMPLICIT NONE
  PUBLIC
  INTEGER, PARAMETER :: NTREAL = c_double        !< DataTypesModule.F90: the real type
  INTEGER, PARAMETER :: NTCOMPLEX = c_double     !< kind of the complex type
  INTEGER, PARAMETER :: NTLONG = c_long
END MODULE DataTypesModule

!> interface blocks for the C ABI (names and argument order of include/ntpoly_amd*.h)
MODULE NTPolyAMDBindings
  USE, INTRINSIC :: ISO_C_BINDING
  IMPLICIT NONE
  PUBLIC
  INTEGER, PARAMETER :: SIZE_wrp = 12
  INTERFACE
     SUBROUTINE ConstructGlobalProcessGrid_wrp(comm, r, c, s) BIND(C, name="ConstructGlobalProcessGrid_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: comm, r, c, s
     END SUBROUTINE
     SUBROUTINE ConstructGlobalProcessGrid_onlyslice_wrp(comm, s) BIND(C, name="ConstructGlobalProcessGrid_onlyslice_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: comm, s
     END SUBROUTINE
     SUBROUTINE ConstructGlobalProcessGrid_default_wrp(comm) BIND(C, name="ConstructGlobalProcessGrid_default_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: comm
     END SUBROUTINE
     SUBROUTINE DestructGlobalProcessGrid_wrp() BIND(C, name="DestructGlobalProcessGrid_wrp")
     END SUBROUTINE
     FUNCTION GetGlobalIsRoot_wrp() BIND(C, name="GetGlobalIsRoot_wrp") RESULT(v)
       IMPORT; LOGICAL(c_bool) :: v
     END FUNCTION
     FUNCTION GetGlobalNumRows_wrp() BIND(C, name="GetGlobalNumRows_wrp") RESULT(v)
       IMPORT; INTEGER(c_int) :: v
     END FUNCTION
     FUNCTION GetGlobalNumColumns_wrp() BIND(C, name="GetGlobalNumColumns_wrp") RESULT(v)
       IMPORT; INTEGER(c_int) :: v
     END FUNCTION
     FUNCTION GetGlobalNumSlices_wrp() BIND(C, name="GetGlobalNumSlices_wrp") RESULT(v)
       IMPORT; INTEGER(c_int) :: v
     END FUNCTION
     FUNCTION GetGlobalMyRow_wrp() BIND(C, name="GetGlobalMyRow_wrp") RESULT(v)
       IMPORT; INTEGER(c_int) :: v
     END FUNCTION
     FUNCTION GetGlobalMyColumn_wrp() BIND(C, name="GetGlobalMyColumn_wrp") RESULT(v)
       IMPORT; INTEGER(c_int) :: v
     END FUNCTION
     FUNCTION GetGlobalMySlice_wrp() BIND(C, name="GetGlobalMySlice_wrp") RESULT(v)
       IMPORT; INTEGER(c_int) :: v
     END FUNCTION
     SUBROUTINE WriteGlobalProcessGridInfo_wrp() BIND(C, name="WriteGlobalProcessGridInfo_wrp")
     END SUBROUTINE
     SUBROUTINE ConstructEmptyMatrix_ps_wrp(ih, n) BIND(C, name="ConstructEmptyMatrix_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE ConstructMatrixFromMatrixMarket_ps_wrp(ih, name, n) BIND(C, name="ConstructMatrixFromMatrixMarket_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); CHARACTER(kind=c_char), INTENT(IN) :: name(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE ConstructMatrixFromBinary_ps_wrp(ih, name, n) BIND(C, name="ConstructMatrixFromBinary_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); CHARACTER(kind=c_char), INTENT(IN) :: name(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE WriteMatrixToMatrixMarket_ps_wrp(ih, name, n) BIND(C, name="WriteMatrixToMatrixMarket_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); CHARACTER(kind=c_char), INTENT(IN) :: name(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE WriteMatrixToBinary_ps_wrp(ih, name, n) BIND(C, name="WriteMatrixToBinary_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); CHARACTER(kind=c_char), INTENT(IN) :: name(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE DestructMatrix_ps_wrp(ih) BIND(C, name="DestructMatrix_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE CopyMatrix_ps_wrp(a, b) BIND(C, name="CopyMatrix_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*); INTEGER(c_int), INTENT(INOUT) :: b(*)
     END SUBROUTINE
     SUBROUTINE FillMatrixIdentity_ps_wrp(ih) BIND(C, name="FillMatrixIdentity_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE FillMatrixFromTripletList_psr_wrp(ih, tl) BIND(C, name="FillMatrixFromTripletList_psr_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*), tl(*)
     END SUBROUTINE
     SUBROUTINE GetMatrixTripletList_psr_wrp(ih, tl) BIND(C, name="GetMatrixTripletList_psr_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(INOUT) :: tl(*)
     END SUBROUTINE
     SUBROUTINE GetMatrixActualDimension_ps_wrp(ih, n) BIND(C, name="GetMatrixActualDimension_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(OUT) :: n
     END SUBROUTINE
     SUBROUTINE GetMatrixLogicalDimension_ps_wrp(ih, n) BIND(C, name="GetMatrixLogicalDimension_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(OUT) :: n
     END SUBROUTINE
     SUBROUTINE GetMatrixSize_ps_wrp(ih, n) BIND(C, name="GetMatrixSize_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_long), INTENT(OUT) :: n
     END SUBROUTINE
     FUNCTION ntpoly_amd_matrix_is_complex(ih) BIND(C, name="ntpoly_amd_matrix_is_complex") RESULT(v)
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int) :: v
     END FUNCTION
     SUBROUTINE TransposeMatrix_ps_wrp(a, at) BIND(C, name="TransposeMatrix_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*); INTEGER(c_int), INTENT(INOUT) :: at(*)
     END SUBROUTINE
     SUBROUTINE ConjugateMatrix_ps_wrp(a) BIND(C, name="ConjugateMatrix_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: a(*)
     END SUBROUTINE
     SUBROUTINE MatrixMultiply_ps_wrp(a, b, c, alpha, beta, thr, pool) BIND(C, name="MatrixMultiply_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), b(*); INTEGER(c_int), INTENT(INOUT) :: c(*), pool(*)
       REAL(c_double), INTENT(IN) :: alpha, beta, thr
     END SUBROUTINE
     SUBROUTINE IncrementMatrix_ps_wrp(a, b, alpha, thr) BIND(C, name="IncrementMatrix_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*); INTEGER(c_int), INTENT(INOUT) :: b(*); REAL(c_double), INTENT(IN) :: alpha, thr
     END SUBROUTINE
     SUBROUTINE ScaleMatrix_ps_wrp(a, c) BIND(C, name="ScaleMatrix_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: a(*); REAL(c_double), INTENT(IN) :: c
     END SUBROUTINE
     SUBROUTINE DotMatrix_psr_wrp(a, b, v) BIND(C, name="DotMatrix_psr_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), b(*); REAL(c_double), INTENT(OUT) :: v
     END SUBROUTINE
     SUBROUTINE MatrixTrace_ps_wrp(a, v) BIND(C, name="MatrixTrace_ps_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*); REAL(c_double), INTENT(OUT) :: v
     END SUBROUTINE
     FUNCTION MatrixNorm_ps_wrp(a) BIND(C, name="MatrixNorm_ps_wrp") RESULT(v)
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*); REAL(c_double) :: v
     END FUNCTION
     SUBROUTINE ConstructMatrixMemoryPool_p_wrp(ih, mat) BIND(C, name="ConstructMatrixMemoryPool_p_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: mat(*)
     END SUBROUTINE
     SUBROUTINE DestructMatrixMemoryPool_p_wrp(ih) BIND(C, name="DestructMatrixMemoryPool_p_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE GershgorinBounds_wrp(a, mn, mx) BIND(C, name="GershgorinBounds_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*); REAL(c_double), INTENT(OUT) :: mn, mx
     END SUBROUTINE
     SUBROUTINE ConstructTripletList_c_wrp(ih, n) BIND(C, name="ConstructTripletList_c_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE DestructTripletList_c_wrp(ih) BIND(C, name="DestructTripletList_c_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE AppendToTripletList_c_wrp(ih, c, r, re, im) BIND(C, name="AppendToTripletList_c_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: c, r; REAL(c_double), INTENT(IN) :: re, im
     END SUBROUTINE
     SUBROUTINE SetTripletAt_c_wrp(ih, idx, c, r, re, im) BIND(C, name="SetTripletAt_c_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: idx, c, r; REAL(c_double), INTENT(IN) :: re, im
     END SUBROUTINE
     SUBROUTINE GetTripletAt_c_wrp(ih, idx, c, r, re, im) BIND(C, name="GetTripletAt_c_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*), idx; INTEGER(c_int), INTENT(OUT) :: c, r; REAL(c_double), INTENT(OUT) :: re, im
     END SUBROUTINE
     FUNCTION GetTripletListSize_c_wrp(ih) BIND(C, name="GetTripletListSize_c_wrp") RESULT(n)
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int) :: n
     END FUNCTION
     SUBROUTINE FillMatrixFromTripletList_psc_wrp(ih, tl) BIND(C, name="FillMatrixFromTripletList_psc_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*), tl(*)
     END SUBROUTINE
     SUBROUTINE GetMatrixTripletList_psc_wrp(ih, tl) BIND(C, name="GetMatrixTripletList_psc_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(INOUT) :: tl(*)
     END SUBROUTINE
     SUBROUTINE ConstructTripletList_r_wrp(ih, n) BIND(C, name="ConstructTripletList_r_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE DestructTripletList_r_wrp(ih) BIND(C, name="DestructTripletList_r_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE AppendToTripletList_r_wrp(ih, c, r, v) BIND(C, name="AppendToTripletList_r_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: c, r; REAL(c_double), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE SetTripletAt_r_wrp(ih, idx, c, r, v) BIND(C, name="SetTripletAt_r_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: idx, c, r; REAL(c_double), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE GetTripletAt_r_wrp(ih, idx, c, r, v) BIND(C, name="GetTripletAt_r_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*), idx; INTEGER(c_int), INTENT(OUT) :: c, r; REAL(c_double), INTENT(OUT) :: v
     END SUBROUTINE
     FUNCTION GetTripletListSize_r_wrp(ih) BIND(C, name="GetTripletListSize_r_wrp") RESULT(n)
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int) :: n
     END FUNCTION
     SUBROUTINE ConstructDefaultPermutation_wrp(ih, n) BIND(C, name="ConstructDefaultPermutation_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE ConstructReversePermutation_wrp(ih, n) BIND(C, name="ConstructReversePermutation_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE ConstructRandomPermutation_wrp(ih, n) BIND(C, name="ConstructRandomPermutation_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE DestructPermutation_wrp(ih) BIND(C, name="DestructPermutation_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE ConstructSolverParameters_wrp(ih) BIND(C, name="ConstructSolverParameters_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE DestructSolverParameters_wrp(ih) BIND(C, name="DestructSolverParameters_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE SetParametersConvergeDiff_wrp(ih, v) BIND(C, name="SetParametersConvergeDiff_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); REAL(c_double), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE SetParametersThreshold_wrp(ih, v) BIND(C, name="SetParametersThreshold_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); REAL(c_double), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE SetParametersStepThreshold_wrp(ih, v) BIND(C, name="SetParametersStepThreshold_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); REAL(c_double), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE SetParametersMaxIterations_wrp(ih, v) BIND(C, name="SetParametersMaxIterations_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE SetParametersBeVerbose_wrp(ih, v) BIND(C, name="SetParametersBeVerbose_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); LOGICAL(c_bool), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE SetParametersMonitorConvergence_wrp(ih, v) BIND(C, name="SetParametersMonitorConvergence_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); LOGICAL(c_bool), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE SetParametersLoadBalance_wrp(ih, perm) BIND(C, name="SetParametersLoadBalance_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: perm(*)
     END SUBROUTINE
     SUBROUTINE ActivateLogger_wrp(start) BIND(C, name="ActivateLogger_wrp")
       IMPORT; LOGICAL(c_bool), INTENT(IN) :: start
     END SUBROUTINE
     SUBROUTINE ActivateLoggerFile_wrp(start, name, n) BIND(C, name="ActivateLoggerFile_wrp")
       IMPORT; LOGICAL(c_bool), INTENT(IN) :: start; CHARACTER(kind=c_char), INTENT(IN) :: name(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE DeactivateLogger_wrp() BIND(C, name="DeactivateLogger_wrp")
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_log_header(t, n) BIND(C, name="ntpoly_amd_log_header")
       IMPORT; CHARACTER(kind=c_char), INTENT(IN) :: t(*); INTEGER(c_int), INTENT(IN) :: n
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_log_enter() BIND(C, name="ntpoly_amd_log_enter")
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_log_exit() BIND(C, name="ntpoly_amd_log_exit")
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_log_element_string(k, nk, v, nv) BIND(C, name="ntpoly_amd_log_element_string")
       IMPORT; CHARACTER(kind=c_char), INTENT(IN) :: k(*), v(*); INTEGER(c_int), INTENT(IN) :: nk, nv
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_log_element_int(k, nk, v) BIND(C, name="ntpoly_amd_log_element_int")
       IMPORT; CHARACTER(kind=c_char), INTENT(IN) :: k(*); INTEGER(c_int), INTENT(IN) :: nk, v
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_log_element_real(k, nk, v) BIND(C, name="ntpoly_amd_log_element_real")
       IMPORT; CHARACTER(kind=c_char), INTENT(IN) :: k(*); INTEGER(c_int), INTENT(IN) :: nk; REAL(c_double), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_log_element_bool(k, nk, v) BIND(C, name="ntpoly_amd_log_element_bool")
       IMPORT; CHARACTER(kind=c_char), INTENT(IN) :: k(*); INTEGER(c_int), INTENT(IN) :: nk; LOGICAL(c_bool), INTENT(IN) :: v
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_log_list_element(k, nk) BIND(C, name="ntpoly_amd_log_list_element")
       IMPORT; CHARACTER(kind=c_char), INTENT(IN) :: k(*); INTEGER(c_int), INTENT(IN) :: nk
     END SUBROUTINE
     SUBROUTINE DensitySolver_c(h, isq, trace, d, e, mu, sp) BIND(C, name="TRS2_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: h(*), isq(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: d(*)
       REAL(c_double), INTENT(IN) :: trace; REAL(c_double), INTENT(OUT) :: e, mu
     END SUBROUTINE
     SUBROUTINE TRS4_c(h, isq, trace, d, e, mu, sp) BIND(C, name="TRS4_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: h(*), isq(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: d(*)
       REAL(c_double), INTENT(IN) :: trace; REAL(c_double), INTENT(OUT) :: e, mu
     END SUBROUTINE
     SUBROUTINE PM_c(h, isq, trace, d, e, mu, sp) BIND(C, name="PM_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: h(*), isq(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: d(*)
       REAL(c_double), INTENT(IN) :: trace; REAL(c_double), INTENT(OUT) :: e, mu
     END SUBROUTINE
     SUBROUTINE HPCP_c(h, isq, trace, d, e, mu, sp) BIND(C, name="HPCP_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: h(*), isq(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: d(*)
       REAL(c_double), INTENT(IN) :: trace; REAL(c_double), INTENT(OUT) :: e, mu
     END SUBROUTINE
     SUBROUTINE ScaleAndFold_c(h, isq, trace, d, homo, lumo, e, sp) BIND(C, name="ScaleAndFold_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: h(*), isq(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: d(*)
       REAL(c_double), INTENT(IN) :: trace, homo, lumo; REAL(c_double), INTENT(OUT) :: e
     END SUBROUTINE
     SUBROUTINE EnergyDensityMatrix_c(h, d, ed, thr) BIND(C, name="EnergyDensityMatrix_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: h(*), d(*); INTEGER(c_int), INTENT(INOUT) :: ed(*); REAL(c_double), INTENT(IN) :: thr
     END SUBROUTINE
     SUBROUTINE McWeenyStep_c(d, dout, thr) BIND(C, name="McWeenyStep_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: d(*); INTEGER(c_int), INTENT(INOUT) :: dout(*); REAL(c_double), INTENT(IN) :: thr
     END SUBROUTINE
     SUBROUTINE McWeenyStepS_c(d, dout, s, thr) BIND(C, name="McWeenyStepS_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: d(*), s(*); INTEGER(c_int), INTENT(INOUT) :: dout(*); REAL(c_double), INTENT(IN) :: thr
     END SUBROUTINE
     SUBROUTINE MatFun_c(a, b, sp) BIND(C, name="InverseSquareRoot_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: b(*)
     END SUBROUTINE
     SUBROUTINE SquareRoot_c(a, b, sp) BIND(C, name="SquareRoot_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: b(*)
     END SUBROUTINE
     SUBROUTINE SignFunction_c(a, b, sp) BIND(C, name="SignFunction_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: b(*)
     END SUBROUTINE
     SUBROUTINE PolarDecomposition_c(a, u, h, sp) BIND(C, name="PolarDecomposition_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: u(*), h(*)
     END SUBROUTINE
     SUBROUTINE Invert_c(a, b, sp) BIND(C, name="Invert_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: b(*)
     END SUBROUTINE
     SUBROUTINE PseudoInverse_c(a, b, sp) BIND(C, name="PseudoInverse_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), sp(*); INTEGER(c_int), INTENT(INOUT) :: b(*)
     END SUBROUTINE
     SUBROUTINE PowerBounds_c(a, v, sp) BIND(C, name="PowerBounds_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: a(*), sp(*); REAL(c_double), INTENT(OUT) :: v
     END SUBROUTINE
     !! local matrices (Source/C/SMatrix_c.h) and the two entry points behind GatherMatrixToProcess / CommSplitMatrix
     SUBROUTINE ConstructMatrixFromTripletList_lsr_wrp(ih, tl, r, c) BIND(C, name="ConstructMatrixFromTripletList_lsr_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: tl(*), r, c
     END SUBROUTINE
     SUBROUTINE ConstructMatrixFromTripletList_lsc_wrp(ih, tl, r, c) BIND(C, name="ConstructMatrixFromTripletList_lsc_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*); INTEGER(c_int), INTENT(IN) :: tl(*), r, c
     END SUBROUTINE
     SUBROUTINE DestructMatrix_lsr_wrp(ih) BIND(C, name="DestructMatrix_lsr_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE DestructMatrix_lsc_wrp(ih) BIND(C, name="DestructMatrix_lsc_wrp")
       IMPORT; INTEGER(c_int), INTENT(INOUT) :: ih(*)
     END SUBROUTINE
     SUBROUTINE GetMatrixRows_lsr_wrp(ih, n) BIND(C, name="GetMatrixRows_lsr_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(OUT) :: n
     END SUBROUTINE
     SUBROUTINE GetMatrixColumns_lsr_wrp(ih, n) BIND(C, name="GetMatrixColumns_lsr_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(OUT) :: n
     END SUBROUTINE
     SUBROUTINE GetMatrixRows_lsc_wrp(ih, n) BIND(C, name="GetMatrixRows_lsc_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(OUT) :: n
     END SUBROUTINE
     SUBROUTINE GetMatrixColumns_lsc_wrp(ih, n) BIND(C, name="GetMatrixColumns_lsc_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(OUT) :: n
     END SUBROUTINE
     SUBROUTINE MatrixToTripletList_lsr_wrp(ih, tl) BIND(C, name="MatrixToTripletList_lsr_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(INOUT) :: tl(*)
     END SUBROUTINE
     SUBROUTINE MatrixToTripletList_lsc_wrp(ih, tl) BIND(C, name="MatrixToTripletList_lsc_wrp")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(INOUT) :: tl(*)
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_gather_matrix_to_process(ih, il, id) BIND(C, name="ntpoly_amd_gather_matrix_to_process")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*), id; INTEGER(c_int), INTENT(INOUT) :: il(*)
     END SUBROUTINE
     SUBROUTINE ntpoly_amd_comm_split_matrix(ih, is, color, split_slice) BIND(C, name="ntpoly_amd_comm_split_matrix")
       IMPORT; INTEGER(c_int), INTENT(IN) :: ih(*); INTEGER(c_int), INTENT(INOUT) :: is(*); INTEGER(c_int), INTENT(OUT) :: color
       LOGICAL(c_bool), INTENT(OUT) :: split_slice
     END SUBROUTINE
  END INTERFACE
CONTAINS
  !> Fortran string -> (character array, length) as the C ABI takes strings (PSMatrixModule_wrp.F90:73-90)
  PURE FUNCTION cstr(s) RESULT(a)
    CHARACTER(len=*), INTENT(IN) :: s
    CHARACTER(kind=c_char) :: a(MAX(1, LEN_TRIM(s)))
    INTEGER :: i
    a(1) = ' '
    DO i = 1, LEN_TRIM(s)
       a(i) = s(i:i)
    END DO
  END FUNCTION cstr
END MODULE NTPolyAMDBindings

MODULE ProcessGridModule   !< ProcessGridModule.F90:15-56, 130-264 (global grid only)
  USE NTPolyAMDBindings
  IMPLICIT NONE
  PRIVATE
  TYPE, PUBLIC :: ProcessGrid_t   !< the members user code reads (ProcessGridModule.F90:15-56); the engine keeps ONE grid
     INTEGER :: num_process_rows = 1, num_process_columns = 1, num_process_slices = 1
     INTEGER :: my_row = 0, my_column = 0, my_slice = 0
  END TYPE ProcessGrid_t
  TYPE(ProcessGrid_t), PUBLIC, SAVE :: global_grid
  PUBLIC :: ConstructProcessGrid, DestructProcessGrid, IsRoot, WriteProcessGridInfo
  INTERFACE ConstructProcessGrid   !< ProcessGridModule.F90: full grid, or only the number of slices (rest chosen)
     MODULE PROCEDURE ConstructProcessGrid_full
     MODULE PROCEDURE ConstructProcessGrid_onlyslice
  END INTERFACE ConstructProcessGrid
CONTAINS
  SUBROUTINE RefreshGlobalGrid()
    global_grid%num_process_rows = GetGlobalNumRows_wrp()
    global_grid%num_process_columns = GetGlobalNumColumns_wrp()
    global_grid%num_process_slices = GetGlobalNumSlices_wrp()
    global_grid%my_row = GetGlobalMyRow_wrp()
    global_grid%my_column = GetGlobalMyColumn_wrp()
    global_grid%my_slice = GetGlobalMySlice_wrp()
  END SUBROUTINE RefreshGlobalGrid
  SUBROUTINE ConstructProcessGrid_onlyslice(world_comm, process_slices_in, be_verbose_in)
    INTEGER, INTENT(IN) :: world_comm
    INTEGER, INTENT(IN), OPTIONAL :: process_slices_in
    LOGICAL, INTENT(IN), OPTIONAL :: be_verbose_in
    IF (PRESENT(process_slices_in)) THEN
       CALL ConstructGlobalProcessGrid_onlyslice_wrp(INT(world_comm, c_int), INT(process_slices_in, c_int))
    ELSE
       CALL ConstructGlobalProcessGrid_default_wrp(INT(world_comm, c_int))
    END IF
    CALL RefreshGlobalGrid
  END SUBROUTINE ConstructProcessGrid_onlyslice
  SUBROUTINE ConstructProcessGrid_full(world_comm, process_rows, process_columns, process_slices, be_verbose_in)
    INTEGER, INTENT(IN) :: world_comm
    INTEGER, INTENT(IN) :: process_rows, process_columns, process_slices
    LOGICAL, INTENT(IN), OPTIONAL :: be_verbose_in
    CALL ConstructGlobalProcessGrid_wrp(INT(world_comm, c_int), INT(process_rows, c_int), &
         & INT(process_columns, c_int), INT(process_slices, c_int))
    CALL RefreshGlobalGrid
  END SUBROUTINE ConstructProcessGrid_full
  SUBROUTINE DestructProcessGrid()
    CALL DestructGlobalProcessGrid_wrp
  END SUBROUTINE DestructProcessGrid
  FUNCTION IsRoot() RESULT(v)
    LOGICAL :: v
    v = GetGlobalIsRoot_wrp()
  END FUNCTION IsRoot
  SUBROUTINE WriteProcessGridInfo()
    CALL WriteGlobalProcessGridInfo_wrp
  END SUBROUTINE WriteProcessGridInfo
END MODULE ProcessGridModule

MODULE LoggingModule   !< LoggingModule.F90
  USE NTPolyAMDBindings
  USE DataTypesModule, ONLY : NTREAL
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: ActivateLogger, DeactivateLogger, EnterSubLog, ExitSubLog, WriteElement, WriteHeader, WriteListElement
  INTERFACE WriteElement
     MODULE PROCEDURE WriteElement_bool, WriteElement_float, WriteElement_int, WriteElement_string
  END INTERFACE WriteElement
CONTAINS
  SUBROUTINE ActivateLogger(start_document_in, file_name_in)
    LOGICAL, INTENT(IN), OPTIONAL :: start_document_in
    CHARACTER(LEN=*), INTENT(IN), OPTIONAL :: file_name_in
    LOGICAL(c_bool) :: start
    start = .FALSE.
    IF (PRESENT(start_document_in)) start = start_document_in
    IF (PRESENT(file_name_in)) THEN
       CALL ActivateLoggerFile_wrp(start, cstr(file_name_in), INT(LEN_TRIM(file_name_in), c_int))
    ELSE
       CALL ActivateLogger_wrp(start)
    END IF
  END SUBROUTINE ActivateLogger
  SUBROUTINE DeactivateLogger()
    CALL DeactivateLogger_wrp
  END SUBROUTINE DeactivateLogger
  SUBROUTINE EnterSubLog()
    CALL ntpoly_amd_log_enter
  END SUBROUTINE EnterSubLog
  SUBROUTINE ExitSubLog()
    CALL ntpoly_amd_log_exit
  END SUBROUTINE ExitSubLog
  SUBROUTINE WriteHeader(header_value)
    CHARACTER(LEN=*), INTENT(IN) :: header_value
    CALL ntpoly_amd_log_header(cstr(header_value), INT(LEN_TRIM(header_value), c_int))
  END SUBROUTINE WriteHeader
  SUBROUTINE WriteListElement(key)
    CHARACTER(LEN=*), INTENT(IN) :: key
    CALL ntpoly_amd_log_list_element(cstr(key), INT(LEN_TRIM(key), c_int))
  END SUBROUTINE WriteListElement
  SUBROUTINE WriteElement_string(key, VALUE)
    CHARACTER(LEN=*), INTENT(IN) :: key, VALUE
    CALL ntpoly_amd_log_element_string(cstr(key), INT(LEN_TRIM(key), c_int), cstr(VALUE), INT(LEN_TRIM(VALUE), c_int))
  END SUBROUTINE WriteElement_string
  SUBROUTINE WriteElement_int(key, VALUE)
    CHARACTER(LEN=*), INTENT(IN) :: key
    INTEGER, INTENT(IN) :: VALUE
    CALL ntpoly_amd_log_element_int(cstr(key), INT(LEN_TRIM(key), c_int), INT(VALUE, c_int))
  END SUBROUTINE WriteElement_int
  SUBROUTINE WriteElement_float(key, VALUE)
    CHARACTER(LEN=*), INTENT(IN) :: key
    REAL(NTREAL), INTENT(IN) :: VALUE
    CALL ntpoly_amd_log_element_real(cstr(key), INT(LEN_TRIM(key), c_int), VALUE)
  END SUBROUTINE WriteElement_float
  SUBROUTINE WriteElement_bool(key, VALUE)
    CHARACTER(LEN=*), INTENT(IN) :: key
    LOGICAL, INTENT(IN) :: VALUE
    LOGICAL(c_bool) :: v
    v = VALUE
    CALL ntpoly_amd_log_element_bool(cstr(key), INT(LEN_TRIM(key), c_int), v)
  END SUBROUTINE WriteElement_bool
END MODULE LoggingModule

MODULE PermutationModule   !< PermutationModule.F90
  USE NTPolyAMDBindings
  USE ProcessGridModule, ONLY : ProcessGrid_t
  IMPLICIT NONE
  PRIVATE
  TYPE, PUBLIC :: Permutation_t
     INTEGER(c_int) :: ih(SIZE_wrp) = 0
  END TYPE Permutation_t
  PUBLIC :: ConstructDefaultPermutation, ConstructReversePermutation, ConstructRandomPermutation, DestructPermutation
CONTAINS
  SUBROUTINE ConstructDefaultPermutation(this, matrix_dimension)
    TYPE(Permutation_t), INTENT(INOUT) :: this
    INTEGER, INTENT(IN) :: matrix_dimension
    CALL DestructPermutation(this)
    CALL ConstructDefaultPermutation_wrp(this%ih, INT(matrix_dimension, c_int))
  END SUBROUTINE ConstructDefaultPermutation
  SUBROUTINE ConstructReversePermutation(this, matrix_dimension)
    TYPE(Permutation_t), INTENT(INOUT) :: this
    INTEGER, INTENT(IN) :: matrix_dimension
    CALL DestructPermutation(this)
    CALL ConstructReversePermutation_wrp(this%ih, INT(matrix_dimension, c_int))
  END SUBROUTINE ConstructReversePermutation
  SUBROUTINE ConstructRandomPermutation(this, matrix_dimension, process_grid_in)
    TYPE(Permutation_t), INTENT(INOUT) :: this
    INTEGER, INTENT(IN) :: matrix_dimension
    TYPE(ProcessGrid_t), INTENT(INOUT), OPTIONAL :: process_grid_in
    CALL DestructPermutation(this)
    CALL ConstructRandomPermutation_wrp(this%ih, INT(matrix_dimension, c_int))
  END SUBROUTINE ConstructRandomPermutation
  SUBROUTINE DestructPermutation(this)
    TYPE(Permutation_t), INTENT(INOUT) :: this
    IF (ANY(this%ih .NE. 0)) CALL DestructPermutation_wrp(this%ih)
    this%ih = 0
  END SUBROUTINE DestructPermutation
END MODULE PermutationModule

MODULE TripletModule   !< TripletModule.F90: the (column, row, value) record
  USE DataTypesModule, ONLY : NTREAL, NTCOMPLEX
  IMPLICIT NONE
  PRIVATE
  TYPE, PUBLIC :: Triplet_r
     INTEGER :: index_column = 0, index_row = 0
     REAL(NTREAL) :: point_value = 0
  END TYPE Triplet_r
  TYPE, PUBLIC :: Triplet_c
     INTEGER :: index_column = 0, index_row = 0
     COMPLEX(NTCOMPLEX) :: point_value = 0
  END TYPE Triplet_c
END MODULE TripletModule

MODULE MatrixMarketModule   !< MatrixMarketModule.F90:19-28: the symmetry kinds of the file format
  IMPLICIT NONE
  PUBLIC
  ENUM, BIND(c)
     ENUMERATOR :: MM_GENERAL = 1
     ENUMERATOR :: MM_SYMMETRIC = 2
     ENUMERATOR :: MM_SKEW_SYMMETRIC = 3
     ENUMERATOR :: MM_HERMITIAN = 4
  END ENUM
END MODULE MatrixMarketModule

MODULE TripletListModule   !< TripletListModule.F90: lists live in the engine; CurrentSize is mirrored for user code
  USE NTPolyAMDBindings
  USE DataTypesModule, ONLY : NTREAL, NTCOMPLEX
  USE TripletModule, ONLY : Triplet_r, Triplet_c
  USE MatrixMarketModule, ONLY : MM_SYMMETRIC, MM_SKEW_SYMMETRIC, MM_HERMITIAN
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: Triplet_r, Triplet_c
  TYPE, PUBLIC :: TripletList_r
     INTEGER(c_int) :: ih(SIZE_wrp) = 0
     INTEGER :: CurrentSize = 0
  END TYPE TripletList_r
  TYPE, PUBLIC :: TripletList_c
     INTEGER(c_int) :: ih(SIZE_wrp) = 0
     INTEGER :: CurrentSize = 0
  END TYPE TripletList_c
  PUBLIC :: ConstructTripletList, DestructTripletList, AppendToTripletList, GetTripletAt, GetTripletListSize, &
       & SetTripletAt, SymmetrizeTripletList, SyncTripletListSize
  INTERFACE ConstructTripletList
     MODULE PROCEDURE ConstructTripletList_r, ConstructTripletList_c
  END INTERFACE ConstructTripletList
  INTERFACE DestructTripletList
     MODULE PROCEDURE DestructTripletList_r, DestructTripletList_c
  END INTERFACE DestructTripletList
  INTERFACE AppendToTripletList
     MODULE PROCEDURE AppendToTripletList_r, AppendToTripletList_c
  END INTERFACE AppendToTripletList
  INTERFACE GetTripletAt
     MODULE PROCEDURE GetTripletAt_r, GetTripletAt_c
  END INTERFACE GetTripletAt
  INTERFACE SetTripletAt
     MODULE PROCEDURE SetTripletAt_r, SetTripletAt_c
  END INTERFACE SetTripletAt
  INTERFACE GetTripletListSize
     MODULE PROCEDURE GetTripletListSize_r, GetTripletListSize_c
  END INTERFACE GetTripletListSize
  INTERFACE SymmetrizeTripletList
     MODULE PROCEDURE SymmetrizeTripletList_r, SymmetrizeTripletList_c
  END INTERFACE SymmetrizeTripletList
  INTERFACE SyncTripletListSize   !< after the engine filled a list (GetMatrixTripletList)
     MODULE PROCEDURE SyncTripletListSize_r, SyncTripletListSize_c
  END INTERFACE SyncTripletListSize
CONTAINS
  SUBROUTINE ConstructTripletList_r(this, size_in)
    TYPE(TripletList_r), INTENT(INOUT) :: this
    INTEGER, INTENT(IN), OPTIONAL :: size_in
    INTEGER(c_int) :: n
    CALL DestructTripletList_r(this)
    n = 0
    IF (PRESENT(size_in)) n = size_in
    CALL ConstructTripletList_r_wrp(this%ih, n)
    this%CurrentSize = n
  END SUBROUTINE ConstructTripletList_r
  SUBROUTINE ConstructTripletList_c(this, size_in)
    TYPE(TripletList_c), INTENT(INOUT) :: this
    INTEGER, INTENT(IN), OPTIONAL :: size_in
    INTEGER(c_int) :: n
    CALL DestructTripletList_c(this)
    n = 0
    IF (PRESENT(size_in)) n = size_in
    CALL ConstructTripletList_c_wrp(this%ih, n)
    this%CurrentSize = n
  END SUBROUTINE ConstructTripletList_c
  SUBROUTINE DestructTripletList_r(this)
    TYPE(TripletList_r), INTENT(INOUT) :: this
    IF (ANY(this%ih .NE. 0)) CALL DestructTripletList_r_wrp(this%ih)
    this%ih = 0
    this%CurrentSize = 0
  END SUBROUTINE DestructTripletList_r
  SUBROUTINE DestructTripletList_c(this)
    TYPE(TripletList_c), INTENT(INOUT) :: this
    IF (ANY(this%ih .NE. 0)) CALL DestructTripletList_c_wrp(this%ih)
    this%ih = 0
    this%CurrentSize = 0
  END SUBROUTINE DestructTripletList_c
  SUBROUTINE AppendToTripletList_r(this, triplet)
    TYPE(TripletList_r), INTENT(INOUT) :: this
    TYPE(Triplet_r), INTENT(IN) :: triplet
    IF (ALL(this%ih .EQ. 0)) CALL ConstructTripletList_r_wrp(this%ih, 0_c_int)
    CALL AppendToTripletList_r_wrp(this%ih, INT(triplet%index_column, c_int), INT(triplet%index_row, c_int), triplet%point_value)
    this%CurrentSize = this%CurrentSize + 1
  END SUBROUTINE AppendToTripletList_r
  SUBROUTINE AppendToTripletList_c(this, triplet)
    TYPE(TripletList_c), INTENT(INOUT) :: this
    TYPE(Triplet_c), INTENT(IN) :: triplet
    IF (ALL(this%ih .EQ. 0)) CALL ConstructTripletList_c_wrp(this%ih, 0_c_int)
    CALL AppendToTripletList_c_wrp(this%ih, INT(triplet%index_column, c_int), INT(triplet%index_row, c_int), &
         & REAL(triplet%point_value, c_double), REAL(AIMAG(triplet%point_value), c_double))
    this%CurrentSize = this%CurrentSize + 1
  END SUBROUTINE AppendToTripletList_c
  SUBROUTINE SetTripletAt_r(this, index, triplet)
    TYPE(TripletList_r), INTENT(INOUT) :: this
    INTEGER, INTENT(IN) :: index
    TYPE(Triplet_r), INTENT(IN) :: triplet
    CALL SetTripletAt_r_wrp(this%ih, INT(index, c_int), INT(triplet%index_column, c_int), &
         & INT(triplet%index_row, c_int), triplet%point_value)
  END SUBROUTINE SetTripletAt_r
  SUBROUTINE SetTripletAt_c(this, index, triplet)
    TYPE(TripletList_c), INTENT(INOUT) :: this
    INTEGER, INTENT(IN) :: index
    TYPE(Triplet_c), INTENT(IN) :: triplet
    CALL SetTripletAt_c_wrp(this%ih, INT(index, c_int), INT(triplet%index_column, c_int), &
         & INT(triplet%index_row, c_int), REAL(triplet%point_value, c_double), REAL(AIMAG(triplet%point_value), c_double))
  END SUBROUTINE SetTripletAt_c
  SUBROUTINE GetTripletAt_r(this, index, triplet)
    TYPE(TripletList_r), INTENT(IN) :: this
    INTEGER, INTENT(IN) :: index
    TYPE(Triplet_r), INTENT(OUT) :: triplet
    INTEGER(c_int) :: c, r
    CALL GetTripletAt_r_wrp(this%ih, INT(index, c_int), c, r, triplet%point_value)
    triplet%index_column = c
    triplet%index_row = r
  END SUBROUTINE GetTripletAt_r
  SUBROUTINE GetTripletAt_c(this, index, triplet)
    TYPE(TripletList_c), INTENT(IN) :: this
    INTEGER, INTENT(IN) :: index
    TYPE(Triplet_c), INTENT(OUT) :: triplet
    INTEGER(c_int) :: c, r
    REAL(c_double) :: re, im
    CALL GetTripletAt_c_wrp(this%ih, INT(index, c_int), c, r, re, im)
    triplet%index_column = c
    triplet%index_row = r
    triplet%point_value = CMPLX(re, im, KIND=NTCOMPLEX)
  END SUBROUTINE GetTripletAt_c
  FUNCTION GetTripletListSize_r(this) RESULT(n)
    TYPE(TripletList_r), INTENT(IN) :: this
    INTEGER :: n
    n = GetTripletListSize_r_wrp(this%ih)
  END FUNCTION GetTripletListSize_r
  FUNCTION GetTripletListSize_c(this) RESULT(n)
    TYPE(TripletList_c), INTENT(IN) :: this
    INTEGER :: n
    n = GetTripletListSize_c_wrp(this%ih)
  END FUNCTION GetTripletListSize_c
  SUBROUTINE SyncTripletListSize_r(this)
    TYPE(TripletList_r), INTENT(INOUT) :: this
    this%CurrentSize = GetTripletListSize_r_wrp(this%ih)
  END SUBROUTINE SyncTripletListSize_r
  SUBROUTINE SyncTripletListSize_c(this)
    TYPE(TripletList_c), INTENT(INOUT) :: this
    this%CurrentSize = GetTripletListSize_c_wrp(this%ih)
  END SUBROUTINE SyncTripletListSize_c
  !> TripletListModule.F90:509-575: append the mirrored off-diagonal entries
  SUBROUTINE SymmetrizeTripletList_r(triplet_list, pattern_type)
    TYPE(TripletList_r), INTENT(INOUT) :: triplet_list
    INTEGER, INTENT(IN) :: pattern_type
    TYPE(Triplet_r) :: trip, trip_t
    INTEGER :: II, initial_size
    initial_size = triplet_list%CurrentSize
    IF (pattern_type .NE. MM_SYMMETRIC .AND. pattern_type .NE. MM_SKEW_SYMMETRIC) RETURN
    DO II = 1, initial_size
       CALL GetTripletAt_r(triplet_list, II, trip)
       IF (trip%index_column .NE. trip%index_row) THEN
          trip_t%index_row = trip%index_column
          trip_t%index_column = trip%index_row
          trip_t%point_value = trip%point_value
          IF (pattern_type .EQ. MM_SKEW_SYMMETRIC) trip_t%point_value = -1.0 * trip%point_value
          CALL AppendToTripletList_r(triplet_list, trip_t)
       END IF
    END DO
  END SUBROUTINE SymmetrizeTripletList_r
  SUBROUTINE SymmetrizeTripletList_c(triplet_list, pattern_type)
    TYPE(TripletList_c), INTENT(INOUT) :: triplet_list
    INTEGER, INTENT(IN) :: pattern_type
    TYPE(Triplet_c) :: trip, trip_t
    INTEGER :: II, initial_size
    initial_size = triplet_list%CurrentSize
    IF (pattern_type .NE. MM_SYMMETRIC .AND. pattern_type .NE. MM_SKEW_SYMMETRIC .AND. pattern_type .NE. MM_HERMITIAN) RETURN
    DO II = 1, initial_size
       CALL GetTripletAt_c(triplet_list, II, trip)
       IF (trip%index_column .NE. trip%index_row) THEN
          trip_t%index_row = trip%index_column
          trip_t%index_column = trip%index_row
          trip_t%point_value = trip%point_value
          IF (pattern_type .EQ. MM_SKEW_SYMMETRIC) trip_t%point_value = -1.0 * trip%point_value
          IF (pattern_type .EQ. MM_HERMITIAN) trip_t%point_value = CONJG(trip%point_value)
          CALL AppendToTripletList_c(triplet_list, trip_t)
       END IF
    END DO
  END SUBROUTINE SymmetrizeTripletList_c
END MODULE TripletListModule

MODULE SMatrixModule   !< SMatrixModule.F90:15-30: local matrices (Matrix_lsr / Matrix_lsc), what GatherMatrixToProcess hands back
  USE NTPolyAMDBindings
  USE TripletListModule, ONLY : TripletList_r, TripletList_c, SyncTripletListSize
  IMPLICIT NONE
  PRIVATE
  TYPE, PUBLIC :: Matrix_lsr
     INTEGER(c_int) :: ih(SIZE_wrp) = 0   !< opaque handle of the engine's local matrix
     INTEGER :: rows = 0, columns = 0
  END TYPE Matrix_lsr
  TYPE, PUBLIC :: Matrix_lsc
     INTEGER(c_int) :: ih(SIZE_wrp) = 0
     INTEGER :: rows = 0, columns = 0
  END TYPE Matrix_lsc
  PUBLIC :: ConstructMatrixFromTripletList, DestructMatrix, GetMatrixRows, GetMatrixColumns, MatrixToTripletList, RefreshLocalMatrix
  INTERFACE ConstructMatrixFromTripletList
     MODULE PROCEDURE ConstructMatrixFromTripletList_lsr, ConstructMatrixFromTripletList_lsc
  END INTERFACE ConstructMatrixFromTripletList
  INTERFACE DestructMatrix
     MODULE PROCEDURE DestructMatrix_lsr, DestructMatrix_lsc
  END INTERFACE DestructMatrix
  INTERFACE GetMatrixRows
     MODULE PROCEDURE GetMatrixRows_lsr, GetMatrixRows_lsc
  END INTERFACE GetMatrixRows
  INTERFACE GetMatrixColumns
     MODULE PROCEDURE GetMatrixColumns_lsr, GetMatrixColumns_lsc
  END INTERFACE GetMatrixColumns
  INTERFACE MatrixToTripletList
     MODULE PROCEDURE MatrixToTripletList_lsr, MatrixToTripletList_lsc
  END INTERFACE MatrixToTripletList
  INTERFACE RefreshLocalMatrix   !< mirror the engine-side shape into the Fortran-visible members
     MODULE PROCEDURE RefreshLocalMatrix_lsr, RefreshLocalMatrix_lsc
  END INTERFACE RefreshLocalMatrix
CONTAINS
  SUBROUTINE RefreshLocalMatrix_lsr(this)
    TYPE(Matrix_lsr), INTENT(INOUT) :: this
    INTEGER(c_int) :: n
    IF (ALL(this%ih .EQ. 0)) RETURN
    CALL GetMatrixRows_lsr_wrp(this%ih, n); this%rows = n
    CALL GetMatrixColumns_lsr_wrp(this%ih, n); this%columns = n
  END SUBROUTINE RefreshLocalMatrix_lsr
  SUBROUTINE RefreshLocalMatrix_lsc(this)
    TYPE(Matrix_lsc), INTENT(INOUT) :: this
    INTEGER(c_int) :: n
    IF (ALL(this%ih .EQ. 0)) RETURN
    CALL GetMatrixRows_lsc_wrp(this%ih, n); this%rows = n
    CALL GetMatrixColumns_lsc_wrp(this%ih, n); this%columns = n
  END SUBROUTINE RefreshLocalMatrix_lsc
  SUBROUTINE ConstructMatrixFromTripletList_lsr(this, triplet_list, rows, columns)   !< SMatrixModule.F90 ConstructMatrixFromTripletList
    TYPE(Matrix_lsr), INTENT(INOUT) :: this
    TYPE(TripletList_r), INTENT(IN) :: triplet_list
    INTEGER, INTENT(IN) :: rows, columns
    CALL DestructMatrix_lsr(this)
    CALL ConstructMatrixFromTripletList_lsr_wrp(this%ih, triplet_list%ih, INT(rows, c_int), INT(columns, c_int))
    CALL RefreshLocalMatrix_lsr(this)
  END SUBROUTINE ConstructMatrixFromTripletList_lsr
  SUBROUTINE ConstructMatrixFromTripletList_lsc(this, triplet_list, rows, columns)
    TYPE(Matrix_lsc), INTENT(INOUT) :: this
    TYPE(TripletList_c), INTENT(IN) :: triplet_list
    INTEGER, INTENT(IN) :: rows, columns
    CALL DestructMatrix_lsc(this)
    CALL ConstructMatrixFromTripletList_lsc_wrp(this%ih, triplet_list%ih, INT(rows, c_int), INT(columns, c_int))
    CALL RefreshLocalMatrix_lsc(this)
  END SUBROUTINE ConstructMatrixFromTripletList_lsc
  SUBROUTINE DestructMatrix_lsr(this)
    TYPE(Matrix_lsr), INTENT(INOUT) :: this
    IF (ANY(this%ih .NE. 0)) CALL DestructMatrix_lsr_wrp(this%ih)
    this%ih = 0; this%rows = 0; this%columns = 0
  END SUBROUTINE DestructMatrix_lsr
  SUBROUTINE DestructMatrix_lsc(this)
    TYPE(Matrix_lsc), INTENT(INOUT) :: this
    IF (ANY(this%ih .NE. 0)) CALL DestructMatrix_lsc_wrp(this%ih)
    this%ih = 0; this%rows = 0; this%columns = 0
  END SUBROUTINE DestructMatrix_lsc
  PURE FUNCTION GetMatrixRows_lsr(this) RESULT(n)
    TYPE(Matrix_lsr), INTENT(IN) :: this
    INTEGER :: n
    n = this%rows
  END FUNCTION GetMatrixRows_lsr
  PURE FUNCTION GetMatrixRows_lsc(this) RESULT(n)
    TYPE(Matrix_lsc), INTENT(IN) :: this
    INTEGER :: n
    n = this%rows
  END FUNCTION GetMatrixRows_lsc
  PURE FUNCTION GetMatrixColumns_lsr(this) RESULT(n)
    TYPE(Matrix_lsr), INTENT(IN) :: this
    INTEGER :: n
    n = this%columns
  END FUNCTION GetMatrixColumns_lsr
  PURE FUNCTION GetMatrixColumns_lsc(this) RESULT(n)
    TYPE(Matrix_lsc), INTENT(IN) :: this
    INTEGER :: n
    n = this%columns
  END FUNCTION GetMatrixColumns_lsc
  SUBROUTINE MatrixToTripletList_lsr(this, triplet_list)   !< SMatrixModule.F90 MatrixToTripletList
    TYPE(Matrix_lsr), INTENT(IN) :: this
    TYPE(TripletList_r), INTENT(INOUT) :: triplet_list
    IF (ALL(triplet_list%ih .EQ. 0)) CALL ConstructTripletList_r_wrp(triplet_list%ih, 0_c_int)
    CALL MatrixToTripletList_lsr_wrp(this%ih, triplet_list%ih)
    CALL SyncTripletListSize(triplet_list)
  END SUBROUTINE MatrixToTripletList_lsr
  SUBROUTINE MatrixToTripletList_lsc(this, triplet_list)
    TYPE(Matrix_lsc), INTENT(IN) :: this
    TYPE(TripletList_c), INTENT(INOUT) :: triplet_list
    IF (ALL(triplet_list%ih .EQ. 0)) CALL ConstructTripletList_c_wrp(triplet_list%ih, 0_c_int)
    CALL MatrixToTripletList_lsc_wrp(this%ih, triplet_list%ih)
    CALL SyncTripletListSize(triplet_list)
  END SUBROUTINE MatrixToTripletList_lsc
END MODULE SMatrixModule

MODULE PSMatrixModule   !< PSMatrixModule.F90:33-51 and the routines user code calls
  USE NTPolyAMDBindings
  USE DataTypesModule, ONLY : NTREAL, NTLONG
  USE TripletListModule, ONLY : TripletList_r, TripletList_c, SyncTripletListSize
  USE SMatrixModule, ONLY : Matrix_lsr, Matrix_lsc, DestructLocalMatrix => DestructMatrix, RefreshLocalMatrix
  IMPLICIT NONE
  PRIVATE
  TYPE, PUBLIC :: Matrix_ps
     INTEGER(c_int) :: ih(SIZE_wrp) = 0              !< opaque handle of the engine's matrix (Source/C/Wrapper.h:4)
     INTEGER :: actual_matrix_dimension = 0
     INTEGER :: logical_matrix_dimension = 0
     LOGICAL :: is_complex = .FALSE.
  END TYPE Matrix_ps
  PUBLIC :: ConstructEmptyMatrix, ConstructMatrixFromMatrixMarket, ConstructMatrixFromBinary, &
       & WriteMatrixToMatrixMarket, WriteMatrixToBinary, DestructMatrix, CopyMatrix, FillMatrixIdentity, &
       & FillMatrixFromTripletList, GetMatrixTripletList, GetMatrixSize, GetMatrixActualDimension, &
       & GetMatrixLogicalDimension, TransposeMatrix, ConjugateMatrix, PrepareOutput, RefreshMatrix, PrintMatrix, &
       & GatherMatrixToProcess, CommSplitMatrix
  INTERFACE GatherMatrixToProcess   !< PSMatrixModule.F90:160-165
     MODULE PROCEDURE GatherMatrixToProcess_psr_id, GatherMatrixToProcess_psr_all, GatherMatrixToProcess_psc_id, GatherMatrixToProcess_psc_all
  END INTERFACE GatherMatrixToProcess
  INTERFACE FillMatrixFromTripletList
     MODULE PROCEDURE FillMatrixFromTripletList_r, FillMatrixFromTripletList_c
  END INTERFACE FillMatrixFromTripletList
  INTERFACE GetMatrixTripletList
     MODULE PROCEDURE GetMatrixTripletList_r, GetMatrixTripletList_c
  END INTERFACE GetMatrixTripletList
  INTERFACE ConstructEmptyMatrix
     MODULE PROCEDURE ConstructEmptyMatrix_dim, ConstructEmptyMatrix_like
  END INTERFACE ConstructEmptyMatrix
CONTAINS
  !> mirror the dimensions / type of the engine-side matrix into the Fortran-visible members
  SUBROUTINE RefreshMatrix(this)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    INTEGER(c_int) :: n
    CALL GetMatrixActualDimension_ps_wrp(this%ih, n)
    this%actual_matrix_dimension = n
    CALL GetMatrixLogicalDimension_ps_wrp(this%ih, n)
    this%logical_matrix_dimension = n
    this%is_complex = ntpoly_amd_matrix_is_complex(this%ih) .NE. 0
  END SUBROUTINE RefreshMatrix
  !> an output argument may arrive unconstructed (the reference constructs it inside): give it a handle
  SUBROUTINE PrepareOutput(this, like)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    TYPE(Matrix_ps), INTENT(IN) :: like
    IF (ALL(this%ih .EQ. 0)) CALL ConstructEmptyMatrix_ps_wrp(this%ih, INT(like%actual_matrix_dimension, c_int))
  END SUBROUTINE PrepareOutput
  SUBROUTINE ConstructEmptyMatrix_dim(this, matrix_dim_, process_grid_in, is_complex_in)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    INTEGER, INTENT(IN) :: matrix_dim_
    INTEGER, INTENT(IN), OPTIONAL :: process_grid_in
    LOGICAL, INTENT(IN), OPTIONAL :: is_complex_in
    CALL DestructMatrix(this)
    CALL ConstructEmptyMatrix_ps_wrp(this%ih, INT(matrix_dim_, c_int))
    CALL RefreshMatrix(this)
  END SUBROUTINE ConstructEmptyMatrix_dim
  SUBROUTINE ConstructEmptyMatrix_like(this, reference_matrix)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    TYPE(Matrix_ps), INTENT(IN) :: reference_matrix
    CALL DestructMatrix(this)
    CALL ConstructEmptyMatrix_ps_wrp(this%ih, INT(reference_matrix%actual_matrix_dimension, c_int))
    CALL RefreshMatrix(this)
  END SUBROUTINE ConstructEmptyMatrix_like
  SUBROUTINE ConstructMatrixFromMatrixMarket(this, file_name, process_grid_in)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    CHARACTER(len=*), INTENT(IN) :: file_name
    INTEGER, INTENT(IN), OPTIONAL :: process_grid_in
    CALL DestructMatrix(this)
    CALL ConstructMatrixFromMatrixMarket_ps_wrp(this%ih, cstr(file_name), INT(LEN_TRIM(file_name), c_int))
    CALL RefreshMatrix(this)
  END SUBROUTINE ConstructMatrixFromMatrixMarket
  SUBROUTINE ConstructMatrixFromBinary(this, file_name, process_grid_in)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    CHARACTER(len=*), INTENT(IN) :: file_name
    INTEGER, INTENT(IN), OPTIONAL :: process_grid_in
    CALL DestructMatrix(this)
    CALL ConstructMatrixFromBinary_ps_wrp(this%ih, cstr(file_name), INT(LEN_TRIM(file_name), c_int))
    CALL RefreshMatrix(this)
  END SUBROUTINE ConstructMatrixFromBinary
  SUBROUTINE WriteMatrixToMatrixMarket(this, file_name)
    TYPE(Matrix_ps), INTENT(IN) :: this
    CHARACTER(len=*), INTENT(IN) :: file_name
    CALL WriteMatrixToMatrixMarket_ps_wrp(this%ih, cstr(file_name), INT(LEN_TRIM(file_name), c_int))
  END SUBROUTINE WriteMatrixToMatrixMarket
  SUBROUTINE WriteMatrixToBinary(this, file_name)
    TYPE(Matrix_ps), INTENT(IN) :: this
    CHARACTER(len=*), INTENT(IN) :: file_name
    CALL WriteMatrixToBinary_ps_wrp(this%ih, cstr(file_name), INT(LEN_TRIM(file_name), c_int))
  END SUBROUTINE WriteMatrixToBinary
  SUBROUTINE DestructMatrix(this)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    IF (ANY(this%ih .NE. 0)) CALL DestructMatrix_ps_wrp(this%ih)
    this%ih = 0
    this%actual_matrix_dimension = 0
    this%logical_matrix_dimension = 0
    this%is_complex = .FALSE.
  END SUBROUTINE DestructMatrix
  SUBROUTINE CopyMatrix(matA, matB)
    TYPE(Matrix_ps), INTENT(IN) :: matA
    TYPE(Matrix_ps), INTENT(INOUT) :: matB
    CALL PrepareOutput(matB, matA)
    CALL CopyMatrix_ps_wrp(matA%ih, matB%ih)
    CALL RefreshMatrix(matB)
  END SUBROUTINE CopyMatrix
  SUBROUTINE FillMatrixIdentity(this)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    CALL FillMatrixIdentity_ps_wrp(this%ih)
  END SUBROUTINE FillMatrixIdentity
  SUBROUTINE FillMatrixFromTripletList_r(this, triplet_list, preduplicated_in, prepartitioned_in)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    TYPE(TripletList_r), INTENT(IN) :: triplet_list
    LOGICAL, INTENT(IN), OPTIONAL :: preduplicated_in, prepartitioned_in
    CALL FillMatrixFromTripletList_psr_wrp(this%ih, triplet_list%ih)
    CALL RefreshMatrix(this)
  END SUBROUTINE FillMatrixFromTripletList_r
  SUBROUTINE FillMatrixFromTripletList_c(this, triplet_list, preduplicated_in, prepartitioned_in)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    TYPE(TripletList_c), INTENT(IN) :: triplet_list
    LOGICAL, INTENT(IN), OPTIONAL :: preduplicated_in, prepartitioned_in
    CALL FillMatrixFromTripletList_psc_wrp(this%ih, triplet_list%ih)
    CALL RefreshMatrix(this)
  END SUBROUTINE FillMatrixFromTripletList_c
  SUBROUTINE GetMatrixTripletList_r(this, triplet_list)
    TYPE(Matrix_ps), INTENT(IN) :: this
    TYPE(TripletList_r), INTENT(INOUT) :: triplet_list
    IF (ALL(triplet_list%ih .EQ. 0)) CALL ConstructTripletList_r_wrp(triplet_list%ih, 0_c_int)
    CALL GetMatrixTripletList_psr_wrp(this%ih, triplet_list%ih)
    CALL SyncTripletListSize(triplet_list)
  END SUBROUTINE GetMatrixTripletList_r
  SUBROUTINE GetMatrixTripletList_c(this, triplet_list)
    TYPE(Matrix_ps), INTENT(IN) :: this
    TYPE(TripletList_c), INTENT(INOUT) :: triplet_list
    IF (ALL(triplet_list%ih .EQ. 0)) CALL ConstructTripletList_c_wrp(triplet_list%ih, 0_c_int)
    CALL GetMatrixTripletList_psc_wrp(this%ih, triplet_list%ih)
    CALL SyncTripletListSize(triplet_list)
  END SUBROUTINE GetMatrixTripletList_c
  !> GatherMatrixToProcess (PSMatrixModule.F90:1704-1808): the whole matrix as a local matrix on the process with this rank
  !> inside its slice (the other processes' local_mat is left alone) ...
  SUBROUTINE GatherMatrixToProcess_psr_id(this, local_mat, within_slice_id)
    TYPE(Matrix_ps), INTENT(IN) :: this
    TYPE(Matrix_lsr), INTENT(INOUT) :: local_mat
    INTEGER, INTENT(IN) :: within_slice_id
    INTEGER(c_int) :: ihl(SIZE_wrp)
    ihl = 0
    CALL ntpoly_amd_gather_matrix_to_process(this%ih, ihl, INT(within_slice_id, c_int))
    IF (ANY(ihl .NE. 0)) THEN
       CALL DestructLocalMatrix(local_mat)
       local_mat%ih = ihl
       CALL RefreshLocalMatrix(local_mat)
    END IF
  END SUBROUTINE GatherMatrixToProcess_psr_id
  !> ... or on every process
  SUBROUTINE GatherMatrixToProcess_psr_all(this, local_mat)
    TYPE(Matrix_ps), INTENT(IN) :: this
    TYPE(Matrix_lsr), INTENT(INOUT) :: local_mat
    CALL GatherMatrixToProcess_psr_id(this, local_mat, -1)
  END SUBROUTINE GatherMatrixToProcess_psr_all
  SUBROUTINE GatherMatrixToProcess_psc_id(this, local_mat, within_slice_id)
    TYPE(Matrix_ps), INTENT(IN) :: this
    TYPE(Matrix_lsc), INTENT(INOUT) :: local_mat
    INTEGER, INTENT(IN) :: within_slice_id
    INTEGER(c_int) :: ihl(SIZE_wrp)
    ihl = 0
    CALL ntpoly_amd_gather_matrix_to_process(this%ih, ihl, INT(within_slice_id, c_int))
    IF (ANY(ihl .NE. 0)) THEN
       CALL DestructLocalMatrix(local_mat)
       local_mat%ih = ihl
       CALL RefreshLocalMatrix(local_mat)
    END IF
  END SUBROUTINE GatherMatrixToProcess_psc_id
  SUBROUTINE GatherMatrixToProcess_psc_all(this, local_mat)
    TYPE(Matrix_ps), INTENT(IN) :: this
    TYPE(Matrix_lsc), INTENT(INOUT) :: local_mat
    CALL GatherMatrixToProcess_psc_id(this, local_mat, -1)
  END SUBROUTINE GatherMatrixToProcess_psc_all
  !> CommSplitMatrix (PSMatrixModule.F90:1489-1541): a copy of the matrix hosted on one half of the process grid.  One
  !> process: the copy, colour 0, split along the slices (distributed_includes/CommSplitMatrix.f90:11-14); more: the grid is
  !> split as SplitProcessGrid does (ProcessGridModule.F90:430-515) and the copy lives on the sub-communicator of this
  !> process's half (csrc/psmatrix.cpp ps_comm_split, ncclCommSplit)
  SUBROUTINE CommSplitMatrix(this, split_mat, my_color, split_slice)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    TYPE(Matrix_ps), INTENT(INOUT) :: split_mat
    INTEGER, INTENT(OUT) :: my_color
    LOGICAL, INTENT(OUT) :: split_slice
    INTEGER(c_int) :: color
    LOGICAL(c_bool) :: ss
    CALL DestructMatrix(split_mat)
    CALL ntpoly_amd_comm_split_matrix(this%ih, split_mat%ih, color, ss)
    CALL RefreshMatrix(split_mat)
    my_color = color
    split_slice = ss
  END SUBROUTINE CommSplitMatrix
  !> PrintMatrix (PSMatrixModule.F90:1271-1290): MatrixMarket text to the console, or to a file
  SUBROUTINE PrintMatrix(this, file_name_in)
    TYPE(Matrix_ps) :: this
    CHARACTER(len=*), OPTIONAL, INTENT(IN) :: file_name_in
    IF (PRESENT(file_name_in)) THEN
       CALL WriteMatrixToMatrixMarket_ps_wrp(this%ih, cstr(file_name_in), INT(LEN_TRIM(file_name_in), c_int))
    ELSE
       CALL WriteMatrixToMatrixMarket_ps_wrp(this%ih, cstr("/dev/stdout"), INT(11, c_int))
    END IF
  END SUBROUTINE PrintMatrix
  FUNCTION GetMatrixSize(this) RESULT(total_size)
    TYPE(Matrix_ps), INTENT(IN) :: this
    INTEGER(NTLONG) :: total_size
    CALL GetMatrixSize_ps_wrp(this%ih, total_size)
  END FUNCTION GetMatrixSize
  FUNCTION GetMatrixActualDimension(this) RESULT(n)
    TYPE(Matrix_ps), INTENT(IN) :: this
    INTEGER :: n
    n = this%actual_matrix_dimension
  END FUNCTION GetMatrixActualDimension
  FUNCTION GetMatrixLogicalDimension(this) RESULT(n)
    TYPE(Matrix_ps), INTENT(IN) :: this
    INTEGER :: n
    n = this%logical_matrix_dimension
  END FUNCTION GetMatrixLogicalDimension
  SUBROUTINE TransposeMatrix(AMat, TransMat)
    TYPE(Matrix_ps), INTENT(IN) :: AMat
    TYPE(Matrix_ps), INTENT(INOUT) :: TransMat
    CALL PrepareOutput(TransMat, AMat)
    CALL TransposeMatrix_ps_wrp(AMat%ih, TransMat%ih)
    CALL RefreshMatrix(TransMat)
  END SUBROUTINE TransposeMatrix
  SUBROUTINE ConjugateMatrix(this)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    CALL ConjugateMatrix_ps_wrp(this%ih)
  END SUBROUTINE ConjugateMatrix
END MODULE PSMatrixModule

MODULE PMatrixMemoryPoolModule   !< PMatrixMemoryPoolModule.F90 (accepted and ignored by the engine)
  USE NTPolyAMDBindings
  IMPLICIT NONE
  PRIVATE
  TYPE, PUBLIC :: MatrixMemoryPool_p
     INTEGER(c_int) :: ih(SIZE_wrp) = 0
  END TYPE MatrixMemoryPool_p
  PUBLIC :: DestructMatrixMemoryPool
CONTAINS
  SUBROUTINE DestructMatrixMemoryPool(this)
    TYPE(MatrixMemoryPool_p), INTENT(INOUT) :: this
    IF (ANY(this%ih .NE. 0)) CALL DestructMatrixMemoryPool_p_wrp(this%ih)
    this%ih = 0
  END SUBROUTINE DestructMatrixMemoryPool
END MODULE PMatrixMemoryPoolModule

MODULE PSMatrixAlgebraModule   !< PSMatrixAlgebraModule.F90
  USE NTPolyAMDBindings
  USE DataTypesModule, ONLY : NTREAL
  USE PSMatrixModule, ONLY : Matrix_ps, PrepareOutput, RefreshMatrix
  USE PMatrixMemoryPoolModule, ONLY : MatrixMemoryPool_p
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: MatrixMultiply, IncrementMatrix, ScaleMatrix, DotMatrix, MatrixTrace, MatrixNorm
CONTAINS
  !> C = alpha*A*B + beta*C with threshold (PSMatrixAlgebraModule.F90:108-269; defaults :133-147)
  SUBROUTINE MatrixMultiply(matA, matB, matC, alpha_in, beta_in, threshold_in, memory_pool_in)
    TYPE(Matrix_ps), INTENT(IN) :: matA, matB
    TYPE(Matrix_ps), INTENT(INOUT) :: matC
    REAL(NTREAL), INTENT(IN), OPTIONAL :: alpha_in, beta_in, threshold_in
    TYPE(MatrixMemoryPool_p), INTENT(INOUT), OPTIONAL :: memory_pool_in
    REAL(c_double) :: alpha, beta, threshold
    INTEGER(c_int) :: pool(SIZE_wrp)
    alpha = 1.0_c_double; beta = 0.0_c_double; threshold = 0.0_c_double
    IF (PRESENT(alpha_in)) alpha = alpha_in
    IF (PRESENT(beta_in)) beta = beta_in
    IF (PRESENT(threshold_in)) threshold = threshold_in
    CALL PrepareOutput(matC, matA)
    pool = 0
    CALL ConstructMatrixMemoryPool_p_wrp(pool, matA%ih)
    CALL MatrixMultiply_ps_wrp(matA%ih, matB%ih, matC%ih, alpha, beta, threshold, pool)
    CALL DestructMatrixMemoryPool_p_wrp(pool)
    CALL RefreshMatrix(matC)
  END SUBROUTINE MatrixMultiply
  SUBROUTINE IncrementMatrix(matA, matB, alpha_in, threshold_in)
    TYPE(Matrix_ps), INTENT(IN) :: matA
    TYPE(Matrix_ps), INTENT(INOUT) :: matB
    REAL(NTREAL), INTENT(IN), OPTIONAL :: alpha_in, threshold_in
    REAL(c_double) :: alpha, threshold
    alpha = 1.0_c_double; threshold = 0.0_c_double
    IF (PRESENT(alpha_in)) alpha = alpha_in
    IF (PRESENT(threshold_in)) threshold = threshold_in
    CALL IncrementMatrix_ps_wrp(matA%ih, matB%ih, alpha, threshold)
    CALL RefreshMatrix(matB)
  END SUBROUTINE IncrementMatrix
  SUBROUTINE ScaleMatrix(this, constant)
    TYPE(Matrix_ps), INTENT(INOUT) :: this
    REAL(NTREAL), INTENT(IN) :: constant
    CALL ScaleMatrix_ps_wrp(this%ih, constant)
  END SUBROUTINE ScaleMatrix
  SUBROUTINE DotMatrix(matA, matB, product)
    TYPE(Matrix_ps), INTENT(IN) :: matA, matB
    REAL(NTREAL), INTENT(OUT) :: product
    CALL DotMatrix_psr_wrp(matA%ih, matB%ih, product)
  END SUBROUTINE DotMatrix
  SUBROUTINE MatrixTrace(this, trace_value)
    TYPE(Matrix_ps), INTENT(IN) :: this
    REAL(NTREAL), INTENT(OUT) :: trace_value
    CALL MatrixTrace_ps_wrp(this%ih, trace_value)
  END SUBROUTINE MatrixTrace
  FUNCTION MatrixNorm(this) RESULT(norm_value)
    TYPE(Matrix_ps), INTENT(IN) :: this
    REAL(NTREAL) :: norm_value
    norm_value = MatrixNorm_ps_wrp(this%ih)
  END FUNCTION MatrixNorm
END MODULE PSMatrixAlgebraModule

MODULE SolverParametersModule   !< SolverParametersModule.F90:14-113: a plain Fortran type, as in the reference
  USE NTPolyAMDBindings
  USE DataTypesModule, ONLY : NTREAL
  USE PermutationModule, ONLY : Permutation_t
  IMPLICIT NONE
  PRIVATE
  TYPE, PUBLIC :: SolverParameters_t
     REAL(NTREAL) :: converge_diff = 1e-6_NTREAL
     INTEGER :: max_iterations = 1000
     REAL(NTREAL) :: threshold = 0.0_NTREAL
     LOGICAL :: be_verbose = .FALSE.
     LOGICAL :: do_load_balancing = .FALSE.
     TYPE(Permutation_t) :: BalancePermutation
     REAL(NTREAL) :: step_thresh = 1e-2_NTREAL
     LOGICAL :: monitor_convergence = .TRUE.
  END TYPE SolverParameters_t
  PUBLIC :: ConstructSolverParameters, DestructSolverParameters, CopySolverParameters, MakeParameterHandle, &
       & FreeParameterHandle
CONTAINS
  SUBROUTINE ConstructSolverParameters(this, converge_diff_in, threshold_in, max_iterations_in, be_verbose_in, &
       & BalancePermutation_in, step_thresh_in, monitor_convergence_in)
    TYPE(SolverParameters_t), INTENT(INOUT) :: this
    REAL(NTREAL), INTENT(IN), OPTIONAL :: converge_diff_in, threshold_in, step_thresh_in
    INTEGER, INTENT(IN), OPTIONAL :: max_iterations_in
    LOGICAL, INTENT(IN), OPTIONAL :: be_verbose_in, monitor_convergence_in
    TYPE(Permutation_t), INTENT(IN), OPTIONAL :: BalancePermutation_in
    this%converge_diff = 1e-6_NTREAL; this%threshold = 0.0_NTREAL; this%max_iterations = 1000
    this%be_verbose = .FALSE.; this%do_load_balancing = .FALSE.; this%step_thresh = 1e-2_NTREAL
    this%monitor_convergence = .TRUE.
    IF (PRESENT(converge_diff_in)) this%converge_diff = converge_diff_in
    IF (PRESENT(threshold_in)) this%threshold = threshold_in
    IF (PRESENT(max_iterations_in)) this%max_iterations = max_iterations_in
    IF (PRESENT(be_verbose_in)) this%be_verbose = be_verbose_in
    IF (PRESENT(step_thresh_in)) this%step_thresh = step_thresh_in
    IF (PRESENT(monitor_convergence_in)) this%monitor_convergence = monitor_convergence_in
    IF (PRESENT(BalancePermutation_in)) THEN
       this%do_load_balancing = .TRUE.
       this%BalancePermutation = BalancePermutation_in   ! shares the engine-side permutation (the caller destructs it)
    END IF
  END SUBROUTINE ConstructSolverParameters
  SUBROUTINE CopySolverParameters(paramA, paramB)
    TYPE(SolverParameters_t), INTENT(IN) :: paramA
    TYPE(SolverParameters_t), INTENT(INOUT) :: paramB
    paramB = paramA
  END SUBROUTINE CopySolverParameters
  SUBROUTINE DestructSolverParameters(this)
    TYPE(SolverParameters_t), INTENT(INOUT) :: this
    this%do_load_balancing = .FALSE.
    this%BalancePermutation%ih = 0
  END SUBROUTINE DestructSolverParameters
  !> engine-side parameter object for one solver call
  SUBROUTINE MakeParameterHandle(this, ih)
    TYPE(SolverParameters_t), INTENT(IN) :: this
    INTEGER(c_int), INTENT(OUT) :: ih(SIZE_wrp)
    LOGICAL(c_bool) :: b
    ih = 0
    CALL ConstructSolverParameters_wrp(ih)
    CALL SetParametersConvergeDiff_wrp(ih, this%converge_diff)
    CALL SetParametersThreshold_wrp(ih, this%threshold)
    CALL SetParametersStepThreshold_wrp(ih, this%step_thresh)
    CALL SetParametersMaxIterations_wrp(ih, INT(this%max_iterations, c_int))
    b = this%be_verbose
    CALL SetParametersBeVerbose_wrp(ih, b)
    b = this%monitor_convergence
    CALL SetParametersMonitorConvergence_wrp(ih, b)
    IF (this%do_load_balancing) CALL SetParametersLoadBalance_wrp(ih, this%BalancePermutation%ih)
  END SUBROUTINE MakeParameterHandle
  SUBROUTINE FreeParameterHandle(ih)
    INTEGER(c_int), INTENT(INOUT) :: ih(SIZE_wrp)
    CALL DestructSolverParameters_wrp(ih)
    ih = 0
  END SUBROUTINE FreeParameterHandle
END MODULE SolverParametersModule

MODULE EigenBoundsModule   !< EigenBoundsModule.F90
  USE NTPolyAMDBindings
  USE DataTypesModule, ONLY : NTREAL
  USE PSMatrixModule, ONLY : Matrix_ps
  USE SolverParametersModule, ONLY : SolverParameters_t, MakeParameterHandle, FreeParameterHandle
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: GershgorinBounds, PowerBounds
CONTAINS
  SUBROUTINE GershgorinBounds(this, min_value, max_value)
    TYPE(Matrix_ps), INTENT(IN) :: this
    REAL(NTREAL), INTENT(OUT) :: min_value, max_value
    CALL GershgorinBounds_wrp(this%ih, min_value, max_value)
  END SUBROUTINE GershgorinBounds
  SUBROUTINE PowerBounds(this, max_value, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: this
    REAL(NTREAL), INTENT(OUT) :: max_value
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    TYPE(SolverParameters_t) :: p
    INTEGER(c_int) :: ih(SIZE_wrp)
    IF (PRESENT(solver_parameters_in)) THEN
       p = solver_parameters_in
    ELSE
       p%max_iterations = 10   ! EigenBoundsModule.F90:81-86
    END IF
    CALL MakeParameterHandle(p, ih)
    CALL PowerBounds_c(this%ih, max_value, ih)
    CALL FreeParameterHandle(ih)
  END SUBROUTINE PowerBounds
END MODULE EigenBoundsModule

MODULE DensityMatrixSolversModule   !< DensityMatrixSolversModule.F90
  USE NTPolyAMDBindings
  USE DataTypesModule, ONLY : NTREAL
  USE PSMatrixModule, ONLY : Matrix_ps, PrepareOutput, RefreshMatrix
  USE SolverParametersModule, ONLY : SolverParameters_t, MakeParameterHandle, FreeParameterHandle
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: TRS2, TRS4, PM, HPCP, ScaleAndFold, EnergyDensityMatrix, McWeenyStep
CONTAINS
  SUBROUTINE run_density(which, H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
    INTEGER, INTENT(IN) :: which
    TYPE(Matrix_ps), INTENT(IN) :: H, ISQ
    REAL(NTREAL), INTENT(IN) :: trace
    TYPE(Matrix_ps), INTENT(INOUT) :: K
    REAL(NTREAL), INTENT(OUT), OPTIONAL :: energy_value_out, chemical_potential_out
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    TYPE(SolverParameters_t) :: p
    INTEGER(c_int) :: ih(SIZE_wrp)
    REAL(c_double) :: e, mu
    IF (PRESENT(solver_parameters_in)) p = solver_parameters_in
    CALL MakeParameterHandle(p, ih)
    CALL PrepareOutput(K, H)
    SELECT CASE(which)
    CASE(1)
       CALL DensitySolver_c(H%ih, ISQ%ih, trace, K%ih, e, mu, ih)
    CASE(2)
       CALL TRS4_c(H%ih, ISQ%ih, trace, K%ih, e, mu, ih)
    CASE(3)
       CALL PM_c(H%ih, ISQ%ih, trace, K%ih, e, mu, ih)
    CASE(4)
       CALL HPCP_c(H%ih, ISQ%ih, trace, K%ih, e, mu, ih)
    END SELECT
    CALL FreeParameterHandle(ih)
    CALL RefreshMatrix(K)
    IF (PRESENT(energy_value_out)) energy_value_out = e
    IF (PRESENT(chemical_potential_out)) chemical_potential_out = mu
  END SUBROUTINE run_density
  SUBROUTINE TRS2(H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: H, ISQ
    REAL(NTREAL), INTENT(IN) :: trace
    TYPE(Matrix_ps), INTENT(INOUT) :: K
    REAL(NTREAL), INTENT(OUT), OPTIONAL :: energy_value_out, chemical_potential_out
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    CALL run_density(1, H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
  END SUBROUTINE TRS2
  SUBROUTINE TRS4(H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: H, ISQ
    REAL(NTREAL), INTENT(IN) :: trace
    TYPE(Matrix_ps), INTENT(INOUT) :: K
    REAL(NTREAL), INTENT(OUT), OPTIONAL :: energy_value_out, chemical_potential_out
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    CALL run_density(2, H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
  END SUBROUTINE TRS4
  SUBROUTINE PM(H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: H, ISQ
    REAL(NTREAL), INTENT(IN) :: trace
    TYPE(Matrix_ps), INTENT(INOUT) :: K
    REAL(NTREAL), INTENT(OUT), OPTIONAL :: energy_value_out, chemical_potential_out
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    CALL run_density(3, H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
  END SUBROUTINE PM
  SUBROUTINE HPCP(H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: H, ISQ
    REAL(NTREAL), INTENT(IN) :: trace
    TYPE(Matrix_ps), INTENT(INOUT) :: K
    REAL(NTREAL), INTENT(OUT), OPTIONAL :: energy_value_out, chemical_potential_out
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    CALL run_density(4, H, ISQ, trace, K, energy_value_out, chemical_potential_out, solver_parameters_in)
  END SUBROUTINE HPCP
  SUBROUTINE ScaleAndFold(H, ISQ, trace, K, homo, lumo, energy_value_out, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: H, ISQ
    REAL(NTREAL), INTENT(IN) :: trace, homo, lumo
    TYPE(Matrix_ps), INTENT(INOUT) :: K
    REAL(NTREAL), INTENT(OUT), OPTIONAL :: energy_value_out
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    TYPE(SolverParameters_t) :: p
    INTEGER(c_int) :: ih(SIZE_wrp)
    REAL(c_double) :: e
    IF (PRESENT(solver_parameters_in)) p = solver_parameters_in
    CALL MakeParameterHandle(p, ih)
    CALL PrepareOutput(K, H)
    CALL ScaleAndFold_c(H%ih, ISQ%ih, trace, K%ih, homo, lumo, e, ih)
    CALL FreeParameterHandle(ih)
    CALL RefreshMatrix(K)
    IF (PRESENT(energy_value_out)) energy_value_out = e
  END SUBROUTINE ScaleAndFold
  SUBROUTINE EnergyDensityMatrix(H, D, ED, threshold_in)
    TYPE(Matrix_ps), INTENT(IN) :: H, D
    TYPE(Matrix_ps), INTENT(INOUT) :: ED
    REAL(NTREAL), INTENT(IN), OPTIONAL :: threshold_in
    REAL(c_double) :: thr
    thr = 0.0_c_double
    IF (PRESENT(threshold_in)) thr = threshold_in
    CALL PrepareOutput(ED, H)
    CALL EnergyDensityMatrix_c(H%ih, D%ih, ED%ih, thr)
    CALL RefreshMatrix(ED)
  END SUBROUTINE EnergyDensityMatrix
  SUBROUTINE McWeenyStep(D, DOut, S_in, threshold_in)
    TYPE(Matrix_ps), INTENT(IN) :: D
    TYPE(Matrix_ps), INTENT(INOUT) :: DOut
    TYPE(Matrix_ps), INTENT(IN), OPTIONAL :: S_in
    REAL(NTREAL), INTENT(IN), OPTIONAL :: threshold_in
    REAL(c_double) :: thr
    thr = 0.0_c_double
    IF (PRESENT(threshold_in)) thr = threshold_in
    CALL PrepareOutput(DOut, D)
    IF (PRESENT(S_in)) THEN
       CALL McWeenyStepS_c(D%ih, DOut%ih, S_in%ih, thr)
    ELSE
       CALL McWeenyStep_c(D%ih, DOut%ih, thr)
    END IF
    CALL RefreshMatrix(DOut)
  END SUBROUTINE McWeenyStep
END MODULE DensityMatrixSolversModule

MODULE SquareRootSolversModule   !< SquareRootSolversModule.F90
  USE NTPolyAMDBindings
  USE PSMatrixModule, ONLY : Matrix_ps, PrepareOutput, RefreshMatrix
  USE SolverParametersModule, ONLY : SolverParameters_t, MakeParameterHandle, FreeParameterHandle
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: SquareRoot, InverseSquareRoot
CONTAINS
  SUBROUTINE SquareRoot(InputMat, OutputMat, solver_parameters_in, order_in)
    TYPE(Matrix_ps), INTENT(IN) :: InputMat
    TYPE(Matrix_ps), INTENT(INOUT) :: OutputMat
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    INTEGER, INTENT(IN), OPTIONAL :: order_in
    TYPE(SolverParameters_t) :: p
    INTEGER(c_int) :: ih(SIZE_wrp)
    IF (PRESENT(solver_parameters_in)) p = solver_parameters_in
    CALL MakeParameterHandle(p, ih)
    CALL PrepareOutput(OutputMat, InputMat)
    CALL SquareRoot_c(InputMat%ih, OutputMat%ih, ih)
    CALL FreeParameterHandle(ih)
    CALL RefreshMatrix(OutputMat)
  END SUBROUTINE SquareRoot
  SUBROUTINE InverseSquareRoot(InputMat, OutputMat, solver_parameters_in, order_in)
    TYPE(Matrix_ps), INTENT(IN) :: InputMat
    TYPE(Matrix_ps), INTENT(INOUT) :: OutputMat
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    INTEGER, INTENT(IN), OPTIONAL :: order_in
    TYPE(SolverParameters_t) :: p
    INTEGER(c_int) :: ih(SIZE_wrp)
    IF (PRESENT(solver_parameters_in)) p = solver_parameters_in
    CALL MakeParameterHandle(p, ih)
    CALL PrepareOutput(OutputMat, InputMat)
    CALL MatFun_c(InputMat%ih, OutputMat%ih, ih)
    CALL FreeParameterHandle(ih)
    CALL RefreshMatrix(OutputMat)
  END SUBROUTINE InverseSquareRoot
END MODULE SquareRootSolversModule

MODULE SignSolversModule   !< SignSolversModule.F90
  USE NTPolyAMDBindings
  USE PSMatrixModule, ONLY : Matrix_ps, PrepareOutput, RefreshMatrix
  USE SolverParametersModule, ONLY : SolverParameters_t, MakeParameterHandle, FreeParameterHandle
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: SignFunction, PolarDecomposition
CONTAINS
  SUBROUTINE SignFunction(InMat, OutMat, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: InMat
    TYPE(Matrix_ps), INTENT(INOUT) :: OutMat
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    TYPE(SolverParameters_t) :: p
    INTEGER(c_int) :: ih(SIZE_wrp)
    IF (PRESENT(solver_parameters_in)) p = solver_parameters_in
    CALL MakeParameterHandle(p, ih)
    CALL PrepareOutput(OutMat, InMat)
    CALL SignFunction_c(InMat%ih, OutMat%ih, ih)
    CALL FreeParameterHandle(ih)
    CALL RefreshMatrix(OutMat)
  END SUBROUTINE SignFunction
  SUBROUTINE PolarDecomposition(InMat, Umat, Hmat, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: InMat
    TYPE(Matrix_ps), INTENT(INOUT) :: Umat
    TYPE(Matrix_ps), INTENT(INOUT), OPTIONAL :: Hmat
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    TYPE(SolverParameters_t) :: p
    TYPE(Matrix_ps) :: Htmp
    INTEGER(c_int) :: ih(SIZE_wrp)
    IF (PRESENT(solver_parameters_in)) p = solver_parameters_in
    CALL MakeParameterHandle(p, ih)
    CALL PrepareOutput(Umat, InMat)
    IF (PRESENT(Hmat)) THEN
       CALL PrepareOutput(Hmat, InMat)
       CALL PolarDecomposition_c(InMat%ih, Umat%ih, Hmat%ih, ih)
       CALL RefreshMatrix(Hmat)
    ELSE
       CALL PrepareOutput(Htmp, InMat)
       CALL PolarDecomposition_c(InMat%ih, Umat%ih, Htmp%ih, ih)
       CALL DestructMatrix_ps_wrp(Htmp%ih)
    END IF
    CALL FreeParameterHandle(ih)
    CALL RefreshMatrix(Umat)
  END SUBROUTINE PolarDecomposition
END MODULE SignSolversModule

MODULE InverseSolversModule   !< InverseSolversModule.F90
  USE NTPolyAMDBindings
  USE PSMatrixModule, ONLY : Matrix_ps, PrepareOutput, RefreshMatrix
  USE SolverParametersModule, ONLY : SolverParameters_t, MakeParameterHandle, FreeParameterHandle
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: Invert, PseudoInverse
CONTAINS
  SUBROUTINE Invert(InputMat, OutputMat, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: InputMat
    TYPE(Matrix_ps), INTENT(INOUT) :: OutputMat
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    TYPE(SolverParameters_t) :: p
    INTEGER(c_int) :: ih(SIZE_wrp)
    IF (PRESENT(solver_parameters_in)) p = solver_parameters_in
    CALL MakeParameterHandle(p, ih)
    CALL PrepareOutput(OutputMat, InputMat)
    CALL Invert_c(InputMat%ih, OutputMat%ih, ih)
    CALL FreeParameterHandle(ih)
    CALL RefreshMatrix(OutputMat)
  END SUBROUTINE Invert
  SUBROUTINE PseudoInverse(InputMat, OutputMat, solver_parameters_in)
    TYPE(Matrix_ps), INTENT(IN) :: InputMat
    TYPE(Matrix_ps), INTENT(INOUT) :: OutputMat
    TYPE(SolverParameters_t), INTENT(IN), OPTIONAL :: solver_parameters_in
    TYPE(SolverParameters_t) :: p
    INTEGER(c_int) :: ih(SIZE_wrp)
    IF (PRESENT(solver_parameters_in)) p = solver_parameters_in
    CALL MakeParameterHandle(p, ih)
    CALL PrepareOutput(OutputMat, InputMat)
    CALL PseudoInverse_c(InputMat%ih, OutputMat%ih, ih)
    CALL FreeParameterHandle(ih)
    CALL RefreshMatrix(OutputMat)
  END SUBROUTINE PseudoInverse
END MODULE InverseSolversModule
