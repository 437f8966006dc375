#!/usr/bin/env python3
"""Drop-in for the reference's `python run.py -m <models> [-a arch] [-i in] [-o out] [-cf] [-comp] [-norm] [-no_fp16]` (run.py:318-445) on the
MI355X engine: the command line lives in innfer_amd/run.py."""
import sys

from innfer_amd.run import main

if __name__ == "__main__":
    sys.exit(main())
