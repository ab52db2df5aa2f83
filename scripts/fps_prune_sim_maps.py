from fps_prune_sim import *
n, m = 4096, 1024
for kind, mk in (("sphere", S.sphere_clouds), ("uniform", S.uniform_clouds)):
    cl = mk(2, n, 0)
    for b in range(2):
        x = cl[b].astype(F)
        fps, vs = chain(x, m)
        g = leaves_axis(x)
        T = touched(x, fps, vs, g)
        wave_cur = [((i1*2+i2) + i0) % 4 for (i0,i1,i2), _ in g]
        report("%s%d cur" % (kind,b), T, wave_cur)
        T2, w2 = sub_or(x, fps, vs)
        report("%s%d half-boxes ORed" % (kind,b), T2, w2)
        # best wave assignment by local search on current leaves (balanced 8 per wave)
        import itertools, random
        random.seed(0)
        assign = list(wave_cur)
        Tm = T[1:]
        def cost(a):
            pw = np.zeros((Tm.shape[0],4), int)
            for gI in range(32): pw[:, a[gI]] += Tm[:, gI]
            return pw.max(1).sum()
        best = cost(assign)
        for it in range(3000):
            i, j = random.sample(range(32), 2)
            if assign[i] == assign[j]: continue
            assign[i], assign[j] = assign[j], assign[i]
            c = cost(assign)
            if c <= best: best = c
            else: assign[i], assign[j] = assign[j], assign[i]
        print("   best map by local search: busiest %.3f" % (best / Tm.shape[0]))
