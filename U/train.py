"""python U/train.py [-c BOOL] [-e N] [-n N] [-t DIR] [-d DIR] -- same flags and outputs as the reference's
U/train.py; the model arithmetic runs on the MI355X kernels (cfd_amd)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from cfd_amd import cli  # noqa: E402

if __name__ == "__main__":
    cli.train("U")
