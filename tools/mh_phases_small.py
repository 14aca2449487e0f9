"""tools/mh_phases.py on launches that do not fill the chip at 128 channels: a 128x11x300 strip
(the near part of a tile: ~27 windows per colour) and a 128x33x300 one (an 8x1 tile's far part)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mh_phases  # noqa: E402

for name, shape in (("strip_128x11x300", (128, 11, 300, 11)), ("strip_128x33x300", (128, 33, 300, 11))):
    bench.WORKLOADS[name] = shape
    sys.argv = ["mh_phases.py", "full", name]
    mh_phases.main()
