python3 - <<'P'
import struct,random
random.seed(1)
n=262144
with open('/tmp/in.bin','wb') as f:
    for _ in range(n): f.write(struct.pack('<72I',*[random.getrandbits(29) for _ in range(72)]))
P
for v in w1_e2 w1_e3 w6_e3 w12_e3 w4_e2; do echo $v; tools/mfma_mont/ubench_$v /tmp/in.bin /tmp/o 262144 256 2 | grep "pair"; done
