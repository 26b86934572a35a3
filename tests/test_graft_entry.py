"""__graft_entry__.build() is what the driver runs every round on the CPU-only container: it must succeed here
(hipcc cross-compiles gfx950 without a GPU) -- a stale assertion inside it once went unnoticed for a whole round."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_build_runs_on_the_cpu_container():
    import __graft_entry__ as g
    g.build()
    from robosimgs_amd import _lib
    from oracle import cpu_ref
    assert os.path.exists(_lib.LIB_PATH) and os.path.exists(cpu_ref.LIB)
