import sys, random
sys.path.insert(0, "/root/repo")
import torch
from mpvss_rs_amd import Engine
EB = 256
eng = Engine(0)
rng = random.Random(1)
n = 65536
pk = eng.batch_exp_fixed_base((2).to_bytes(EB, "big"), b"".join(rng.randrange(1, 1 << 2040).to_bytes(EB, "big") for _ in range(n)))
for _ in range(3):
    ks = eng.keyset_create(pk)
    eng.keyset_destroy(ks)
