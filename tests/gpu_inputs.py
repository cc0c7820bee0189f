"""bench/inputs.py (torch-generated random field elements on the device) for the GPU tests."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench"))
from inputs import random_elements  # noqa: E402,F401
