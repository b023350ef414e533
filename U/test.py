"""python U/test.py [processed_folder [train_folder]] -- per-case evaluation of the saved model (reference:
U/test.py; which cases are scored and the de-normalisation: cfd_amd.cli.EVAL_PLANS[("U", "test")])."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from cfd_amd import cli  # noqa: E402

if __name__ == "__main__":
    folder = sys.argv[1] if len(sys.argv) > 1 else None
    train_folder = sys.argv[2] if len(sys.argv) > 2 else None
    cli.evaluate("U", folder, mode="test", train_folder=train_folder)
