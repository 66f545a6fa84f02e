import sys, os, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import oracle as orc, io_formats as iof
from golden_util import load_case, oracle_setup, CASES
import bioem_amd.engine as eng
import ctypes as C
def pd_of(S):
    pd = eng.ParamDevice()
    for f,_ in eng.ParamDevice._fields_: setattr(pd, f, getattr(S.pd, f))
    return pd
for name in (sys.argv[1:] or CASES):
    case=load_case(name); S=oracle_setup(case)
    for algo in case['algos']:
        t0=time.time()
        E=eng.Engine(pd_of(S), S.nMaps, S.nAngles, S.nCTF, algo)
        E.upload_particles(S.refFFT, S.sumRef, S.sumsqRef)
        E.upload_ctf(S.refCTF, S.ctfParam)
        E.upload_model(S.points, S.NormDen, S.px, S.P['shiftX'], S.P['shiftY'])
        E.upload_orientations(S.angles, S.isQuat)
        raw,pmap,pang=eng.new_prob_block(S.nMaps,S.nAngles,S.pd.writeAngles)
        E.start_run(raw); E.project_convolve_compare(0,S.nAngles); E.finish_run(raw)
        t1=time.time(); print('  gpu done %.2fs'%(t1-t0), flush=True)
        om,oa=S.run(algo); print('  oracle done %.2fs'%(time.time()-t1), 'threads', orc.lib().orc_get_max_threads(), flush=True)
        dl=max(abs(S.final_logp(a)-S.final_logp(b)) for a,b in zip(pmap,om))
        same=all((a['cent_x'],a['cent_y'],a['orient'],a['conv'])==(b['cent_x'],b['cent_y'],b['orient'],b['conv']) for a,b in zip(pmap,om))
        print(name,'algo',algo,'fast',E.fast_path,'max|dlogP| vs oracle=%.3e'%dl,'argmax same',same,'gpu %.2fs'%(t1-t0),flush=True)
        if not same:
            for a,b in zip(pmap,om): print('   gpu',a,'\n   orc',b)
        E.close()
