#!/usr/bin/env python3
"""Writes the riskiness score table that SJPEG_YUV_AUTO / SjpegCompress / SjpegRiskiness need.

The table is trained data of the reference (sjpeg::kSharpnessScore, src/score_7.cc) and is not
part of this repository.  Users who have the reference built as a shared library (this repo's
oracle/Makefile builds one from /root/reference: oracle/_ref/libsjpeg_ref.so) extract it once:

    python tools/extract_riskiness_table.py path/to/libsjpeg.so riskiness.bin
    export SJPEG_HIP_RISKINESS_TABLE=$PWD/riskiness.bin       # or sjpeg_hip_set_riskiness_table()
"""
import ctypes
import sys


def main():
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    lib = ctypes.CDLL(sys.argv[1])
    table = bytes((ctypes.c_uint8 * 117649).in_dll(lib, "_ZN5sjpeg15kSharpnessScoreE"))
    with open(sys.argv[2], "wb") as f:
        f.write(table)
    print("wrote", len(table), "bytes to", sys.argv[2])


if __name__ == "__main__":
    main()
