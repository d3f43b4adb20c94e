#!/usr/bin/env python3
"""MI355X drop-in for MedPy's bin/medpy_graphcut_label_w_regional.py: the same command line as
medpy_amd_graphcut_label.py, whose ``--regional atlas --radditional IMAGE --alpha FLOAT`` options it shares."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd.cli.graphcut_label import main  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(main())
