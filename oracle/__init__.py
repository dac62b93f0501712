"""CPU oracle: test infrastructure only.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (fenicsx-beat_amd/beat) never does."""
