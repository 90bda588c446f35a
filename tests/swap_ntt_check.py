import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
from helpers import load_engine
import cases
from fhe_oracle import coracle
from fhe_oracle.rq import Context as OCtx
kind = sys.argv[1]
fhe = load_engine(kind)
n = 8192
mods = [1152921504606830593, 1152921504606748673, 4611686018427322369]
cases.case_ntt(fhe, kind == 'hip', n, moduli=mods, batch=3, coracle_ctx=coracle.CCtx(OCtx(mods, n)))
print("swap NTT ok", kind, os.environ.get("FHE_LAB_NTT_SWAP"))
