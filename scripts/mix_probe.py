import os, sys; sys.path.insert(0,'.')
import reforge_amd as rf
from tests import util
ctx=rf.Context(0)
WIDE = """input -> aa -> m1:input_image0
input -> bb -> m1:input_image1
input -> cc -> m2:input_image0
m1 -> m2:input_image1
m2 -> output
aa: gaussian5 { sigma: 1.0 }
bb: sharpen { amount: 0.5 }
cc: gaussian9 { sigma: 2.0 }
m1: combination { mix: 0.5 }
m2: combination { mix: 0.5 }"""
for serial in ("0", "1"):
  os.environ["RF_CONCURRENT_LAYERS"] = "0" if serial == "1" else "1"
  for W,H in ((1920,1080),(3840,2160)):
   for gname,text in (("diamond",util.DIAMOND),("wide",WIDE)):
    fmt,name=util.F32,"f32 serial=%s %dx%d %s"%(serial,W,H,gname)
    g=rf.Graph(ctx, rf.Config(text), W,H, fmt)
    g.fill_synthetic(3); g.execute(); g.wait()
    print(name, [(l, round(t*1e3,1)) for l,t in g.time_launches(30)], "frame us", round(g.time_frames(30)/30*1e3,1))
    g.close()

