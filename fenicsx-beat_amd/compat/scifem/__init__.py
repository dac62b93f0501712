from beat.grid import evaluate_function  # noqa: F401
