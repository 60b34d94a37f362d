"""tests -> tools/: the development tools are scripts, not a package."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from kernel_resources import kernel_table  # noqa: E402,F401
from kernel_names import timer_name  # noqa: E402,F401
