import sys; sys.path[:0]=['/root/repo','/root/repo/path-tracer_amd/py']
import numpy as np, ptr_amd
from bench import example1_soa
r=ptr_amd.Renderer(0); r.set_scene(example1_soa()); r.set_background([2,2,2.])
for (w,h) in ((640,480),(1920,1080)):
    v=r.render(w,h,flags=ptr_amd.MODE_F32_ONLY|0x8000)
    full=(v>>8)&0xff; pre=(v>>16)&0xff
    print(w,h,'mean full tests per pixel-wave',full.mean(),'mean prefilter calls',pre.mean(), 'max',full.max(),pre.max())
