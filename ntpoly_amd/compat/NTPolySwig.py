"""`import NTPolySwig as nt` for programs written against the reference's SWIG module (Source/Swig/NTPolySwig.i):
put this directory on PYTHONPATH and the names resolve to the MI355X engine's Python mirror (ntpoly_amd.host).

    PYTHONPATH=<repo>:<repo>/ntpoly_amd/compat python main.py ...

The mirror follows the SWIG classes (Matrix_ps, TripletList_r / Triplet_r, SolverParameters, Permutation,
DensityMatrixSolvers, SquareRootSolvers, ...); solver calls return their output scalars (energy, chemical potential)
as the SWIG typemaps do.  MPI is not involved: under a launcher that sets RANK / WORLD_SIZE / LOCAL_RANK the engine
builds its own RCCL communicator (ntpoly_amd.host.init_comm_from_env), otherwise it runs on one GPU."""
from ntpoly_amd.host import *  # noqa: F401,F403
from ntpoly_amd import host as _host

_host.init_comm_from_env()
