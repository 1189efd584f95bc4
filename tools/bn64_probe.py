"""tools/bn64_probe.py -- full products of the trailing updates' shapes at n = 8192 (K = 256) with 128 x 128 tiles against
128 x 64 tiles (GPX_GEMM_BN64_TILES forces the latter): is the smaller tile's round quantisation worth its operand traffic?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench
for M in (7680, 5632, 3584, 2048):
    for N in (M // 2,):                 # a full M x M/2 product has the tile count of the M x M lower triangle
        for K in (256, 512):
            for bn in ("0", "1000000"):
                os.environ["GPX_GEMM_BN64_TILES"] = bn
                print("BN64 forced" if bn != "0" else "BN128      ", end=" ")
                gemm_bench.run(M, N, K, 0, reps=4, lda=8192)
