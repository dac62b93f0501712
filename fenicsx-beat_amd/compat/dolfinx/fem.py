from beat.ecg import assemble_scalar  # noqa: F401
from beat.grid import Constant, Expression, Function, FunctionSpace, functionspace  # noqa: F401


def form(f, **kw):
    """Forms are not symbolic here: lead forms (beat.ecg.LeadForm) pass through unchanged."""
    return f
