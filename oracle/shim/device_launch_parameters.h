/* ORACLE shim: intentionally empty (see cuda_runtime.h in this directory). */
