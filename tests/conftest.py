import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# PyTorch carries a HIP runtime of its own.  When it is loaded first, libatomdev.so binds to it as well (same soname) and
# both share the device; loaded second, it opens the device a second time inside the process and finds none.  Tests that
# use torch.cuda together with the library therefore need torch imported before the library is.
import torch  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
