import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from theia_amd import workloads as S
from theia_amd.hp.pipeline import Pipeline
for n in (1, 64, 256, 1024, 4096, 16384, 65536, 262144, 1000000):
    for mpl in (10, 2):
        sc = S.config2_sphere(batch=n, maxPathLength=mpl)
        pipe = Pipeline(sc.stages()); ms=[]
        for k in range(12):
            pipe.run(k%2); ms.append(sc.tracer.program.last_kernel_ms(k%2))
        scs = S.config2_sphere(batch=n, maxPathLength=mpl, stats=True); S.run_batches(scs,1)
        st = scs.tracer.program.read_stats()
        print(f"n={n:8d} maxPathLength={mpl:2d} kernel ms median={np.median(ms[4:]):.4f} min={min(ms[4:]):.4f} steps/photon={st[19]/n:.2f} maxIter={int(st[9])}", flush=True)
