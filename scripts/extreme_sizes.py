import sys; sys.path.insert(0,'.')
import numpy as np
import reforge_amd as rf
from oracle import pixel
from tests import util
ctx=rf.Context(0)
pixel.set_threads(16)
G15="input -> gg -> output\ngg: gaussian { sigma: 5.0, radius: 15 }"
C31="input -> cc -> output\ncc: conv2d { ksize: 31, sigma: 5.0 }"
for (W,H) in ((65536,8),(3,100000),(1,1),(2,2),(100000,1),(1,100000),(17,65537)):
    for fmt in (util.F32, util.U8):
        x=pixel.fill_synthetic(W,H,fmt,3)
        for name,text in (("chain3",util.CHAIN3),("chain5",util.CHAIN5),("diamond",util.DIAMOND),("g15",G15),("c31",C31)):
            if name=="c31" and W*H>300000: continue
            want=util.run_oracle(text,x)
            for flags in (0, rf.RF_GRAPH_NO_FUSION):
                got=util.run_hip(ctx,text,x,flags=flags)
                ok=got.tobytes()==want.tobytes()
                if not ok: print("MISMATCH",W,H,fmt,name,flags,flush=True)
    print("size",W,H,"done",flush=True)
