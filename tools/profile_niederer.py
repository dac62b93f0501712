"""Where does a small-grid split step spend its host time?  cProfile of the Niederer demo's loop.
usage: python tools/profile_niederer.py [dx [dt [T]]]   (defaults 0.5 0.05 30)"""
import cProfile
import pstats
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "demos"))
dx, dt, T = (sys.argv[1:] + ["0.5", "0.05", "30"][len(sys.argv) - 1:])[:3]
sys.argv = ["niederer_benchmark.py", "--dx", dx, "--dt", dt, "--T", T]
import niederer_benchmark as nb  # noqa: E402

pr = cProfile.Profile()
pr.enable()
nb.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
