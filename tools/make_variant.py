#!/usr/bin/env python3
"""Development aid: variant libraries for A/B runs of resident_async_kernel -- k_resident.hip
compiled with an experiment's macros, linked with the product's other objects (a minute per
variant instead of a full build).
    python tools/make_variant.py '{"name": ["-DMC3HIP_RA_TAKE_AT=4", ...], ...}'
    gpurun -- 'bash tools/run_variants.sh'        # _variants/libmc3hip_<name>.so each"""
import sys, os, subprocess, glob
sys.path.insert(0,'/root/repo')
from mc3_amd import build as b
def mk(name, extra):
    obj=f'/root/repo/build/kres_{name}.o'
    cmd=[b.hipcc()]+b.CFLAGS+['-DMC3HIP_DEV']+list(extra)+['-c', os.path.join(b.CSRC,'k_resident.hip'), '-o', obj]
    subprocess.check_call(cmd)
    objs=['/root/repo/build/obj/'+u[:-4]+'.o' for u in b.UNITS if u!='k_resident.hip']+[obj]
    os.makedirs('/root/repo/_variants', exist_ok=True)
    lib=f'/root/repo/_variants/libmc3hip_{name}.so'
    subprocess.check_call([b.hipcc()]+b.LDFLAGS+['-o', lib]+objs)
    return lib
if __name__=='__main__':
    import json
    V=json.loads(sys.argv[1])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        for r in ex.map(lambda kv: mk(kv[0], kv[1]), V.items()):
            print(r)
