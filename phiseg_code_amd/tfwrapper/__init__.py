"""Drop-in counterpart of the reference's ``tfwrapper`` package (layers / normalisation / utils):
same function names, argument names and defaults, but every call builds nodes of
``phiseg_code_amd.graph`` that lower to hand-written HIP kernels (libphx.so)."""
