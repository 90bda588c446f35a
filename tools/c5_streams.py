import sys; sys.path.insert(0,"tools"); sys.path.insert(0,".")
import json, torch
import fhe_rs_amd as fhe
import bench_configs as b
n, L = 32768, 16
t = fhe.generate_prime(20, 2 * n, 1 << 20)
par = fhe.BfvParameters(n, t, moduli_sizes=[60] * L)
ctx = par.context_at_level(0)
rk = fhe.RelinearizationKey(b.key_for(ctx, 0xF4E50005))
for batch in (64, 64):
    a = ctx.synth_uniform(0xF4E50005, 0, 0, 2, batch); c = ctx.synth_uniform(0xF4E50005, 0, 2, 2, batch)
    for streams, chunk in ((2, 8), (2, 4), (2, 6), (2, 12)):
        m = fhe.Multiplicator.default(par, rk, 0, mod_switch=True).set_streams(streams).set_chunk(chunk)
        ms = b.timeit(lambda: m.multiply(a, c))
        print(json.dumps(dict(batch=batch, streams=streams, chunk=chunk, ms=round(ms,3), ops_per_s=round(batch/ms*1e3,1))))
