#!/usr/bin/env python3
"""Worst error of the AS-norm row statistics against the fp64 oracle for the two forms of the fused cohort GEMM (split: three
bf16 pieces x six passes; fp32-input MFMAs: NPLDA_COHORT_SPLIT=0), same inputs, same process.  usage: cohort_split_accuracy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import nplda_oracle as orc
from tests.test_cohort_fused_gpu import setup

for D, R, M, topn in ((150, 600, 10000, 500), (170, 300, 6000, 200), (150, 256, 20000, 500)):
    ops, packed, zr, qr, zc, qc, C = setup(D, R, M, 11 + D)
    for select in ("lowest", "highest"):
        ref = orc.cohort_stats(C, topn, select)
        scale = np.abs(ref).max(axis=0)
        out = []
        for split in ("0", "1"):
            os.environ["NPLDA_COHORT_SPLIT"] = split
            got = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select=select).cpu().numpy()
            out.append(np.abs(got - ref).max(axis=0))
        del os.environ["NPLDA_COHORT_SPLIT"]
        print(f"D={D} R={R} M={M} top-{topn} {select}: columns (mean, std, top mean, top std), |ref| max " +
              " ".join(f"{v:.3g}" for v in scale))
        print("   max |err| fp32-input MFMA : " + " ".join(f"{v:.2e}" for v in out[0]))
        print("   max |err| split bf16 x 6  : " + " ".join(f"{v:.2e}" for v in out[1]))
