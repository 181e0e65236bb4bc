#!/bin/bash
# upper bound for a row-Winograd F(2,3) trunk: the shipped kernel with one MFMA in three not issued (S2SR_DIAG_SKIPDY2), A/B/A/B on one box
bash tools/ab_macro.sh S2SR_DIAG_SKIPDY2 "0 1" --hp 1
