from beat.grid import COMM_WORLD, Comm  # noqa: F401

SUM = "sum"
Intracomm = Comm
