"""Where does a small-grid split step spend its host time?  cProfile of the Niederer demo's loop (dx = 0.5 mm)."""
import cProfile
import pstats
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "demos"))
sys.argv = ["niederer_benchmark.py", "--dx", "0.5", "--dt", "0.05", "--T", "30"]
import niederer_benchmark as nb  # noqa: E402

pr = cProfile.Profile()
pr.enable()
nb.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
