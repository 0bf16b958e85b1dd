import sys, time, numpy as np
sys.path[:0]=['/root/repo','/root/repo/fusion-sim_amd','/root/repo/tests']
import fusionpic as fp
from helpers import make_spec, uniform_plasma, frame_sink
side=3162
spec=make_spec(1024,1024,side)
n=side*side
pos,vel,ent,rand=uniform_plasma(n,spec,seed=3,v_th=1e-3)
for bits in (0,4,0,4):
    sim=fp.makeCylindricalParticlePusher(spec,raster_subpixel_bits=bits)
    sim.set(position=pos,velocity=vel,sink_mask=frame_sink(1024,1024),source_pdf=frame_sink(1024,1024))
    sim.setRandomState(ent,rand); sim.addBZ(0.01); sim.precalc(); sim.sort()
    for _ in range(3): sim.precalc(); sim.step(); sim.density()
    sim.sync(); sim.resetStats(); sim.profile(True)
    t0=time.perf_counter()
    for _ in range(20): sim.precalc(); sim.step(); sim.density()
    sim.sync(); el=time.perf_counter()-t0
    st=sim.stats()
    print(bits, 'frame ms %.3f'%(1e3*el/20), {k:(round(v,3) if isinstance(v,float) else v) for k,v in st.items() if k in ('ms_push','ms_deposit','ms_stamp','ms_precalc','ms_sort','step_launches','deposit_launches','sort_passes','deposit_spilled')})
    sim.destroy()
