python3 - <<'P'
import struct,random
random.seed(1)
n=262144
with open('/tmp/in.bin','wb') as f:
    for _ in range(n): f.write(struct.pack('<72I',*[random.getrandbits(29) for _ in range(72)]))
P
for w in 2 4 8; do tools/mfma_mont/ubench_w$w /tmp/in.bin /tmp/o 262144 256 2 | grep "pair"; done
