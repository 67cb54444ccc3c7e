"""Host-side scalar helpers of the RT path (the reference's maths/ package surface that
JetModel / Pipeline need).  Grid-sized arithmetic does not live here: it runs in librjprt."""
from . import geometry, physics, rrls  # noqa: F401
