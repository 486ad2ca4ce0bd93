"""dev: ssd_conv_chain at batch 64 with 8 and 4 waves per workgroup"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssd_object_detection_amd import _lib
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_chain.py")).read().split("for i, (d, w, b, o) in enumerate")[0]
for waves in (8, 4):
    _lib.check(_lib.lib().ssd_dev_knob(b"SSD_CHAIN_WAVES", waves))
    print("waves", waves)
    exec(src)
