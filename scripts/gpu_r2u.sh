#!/bin/bash
# round-2 visit U: library GEMM yardstick at the contraction shapes of both hot paths (evidence for DESIGN section 8)
mkdir -p gpurun_out
timeout 300 python scripts/blas_reference.py > gpurun_out/r2u_blas_reference.log 2>&1; cat gpurun_out/r2u_blas_reference.log | tail -14
