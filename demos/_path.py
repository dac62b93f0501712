"""Puts the in-tree package (fenicsx-beat_amd/beat) on sys.path for the demo scripts."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))
