"""Test infrastructure: lets a plain `python bench.py ...` child process (the ranks bench.py spawns itself for `--gpus N`) run on the
emulated kernel library.  Active only when the test put this directory on PYTHONPATH and set NT_EMU / NT_ROOT; imports
tests/emu/emu_plugin.py ("cuda" tensors -> host memory, product loader -> the emulated library) before the script starts."""
import os
import sys

if os.environ.get("NT_EMU") and os.environ.get("NT_ROOT"):
    sys.path[:0] = [os.environ["NT_EMU"], os.environ["NT_ROOT"], os.path.join(os.environ["NT_ROOT"], "tests")]
    import emu_plugin  # noqa: F401
