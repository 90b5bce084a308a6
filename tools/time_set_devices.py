"""where the seconds of awry_set_devices go at GRCh38 scale (AWRY_VERBOSE laps of make_replica), for a freshly built and for a
loaded index.  usage: time_set_devices.py [text_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AWRY_VERBOSE"] = "1"
import numpy as np
from tests import synth
import awry_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_100_000_000
t = time.time(); text, st, hd = synth.make_text(n, 0, 0xA5A50002, 25 if n > 1e9 else 1, 0.05); print("text %.1f s" % (time.time() - t), flush=True)
t = time.time(); ix = awry_amd.FmIndex.from_text(text, 0, 8, 0, st, hd, build_device=0); print("build %.1f s" % (time.time() - t), flush=True)
t = time.time(); ix.set_devices([0]); print("set_devices %.1f s" % (time.time() - t), flush=True)
p = "/dev/shm/awry_t.awry" if os.access("/dev/shm", os.W_OK) else "/tmp/awry_t.awry"
t = time.time(); ix.save(p); print("save %.1f s" % (time.time() - t), flush=True)
ix.close()
t = time.time(); ix2 = awry_amd.FmIndex.load(p); print("load %.1f s" % (time.time() - t), flush=True)
t = time.time(); ix2.set_devices([0]); print("set_devices (loaded) %.1f s" % (time.time() - t), flush=True)
os.remove(p)
