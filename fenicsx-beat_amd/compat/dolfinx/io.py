from beat.grid import VTXWriter  # noqa: F401
