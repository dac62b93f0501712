from beat.grid import COMM_SELF, COMM_WORLD, Comm  # noqa: F401

SUM = "sum"
MAX = "max"
MIN = "min"
PROD = "prod"
Intracomm = Comm
