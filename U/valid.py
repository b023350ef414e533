"""python U/valid.py [processed_folder] -- per-case evaluation of the saved generator (reference: U/valid.py)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from cfd_amd import cli  # noqa: E402

if __name__ == "__main__":
    folder = sys.argv[1] if len(sys.argv) > 1 else "./100_case_si/processed/"
    cli.evaluate("U", folder, out_dir="./U/evaluation_valid/")
