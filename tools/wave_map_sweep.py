"""Build timing variants of the piece kernel that differ only in the wave -> piece-slot / pin-share maps
(HIVE_WAVE_SLOTS / HIVE_PIN_IDS, csrc/hive_env.hip) into build/abl/abl_*.so; tools/ablate.py times them on the GPU.
Candidates come from a small load model (Ant 13, Spider 8.5, light 3.3, pin share 7.3 per cent of a
workgroup's SIMD time; wave w runs on SIMD w mod 4)."""
import itertools, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = {0: 3.3, 1: 3.3, 2: 3.3, 3: 8.5, 4: 8.5, 5: 3.3, 6: 3.3, 7: 3.3, 8: 13, 9: 13, 10: 13}
PIN = 7.3


def hexmap(digits):
    return "0x" + "".join("%X" % d for d in reversed(digits)) + "ull"


def load(slots, pins):
    simd = [0.0] * 4
    for w, (s, p) in enumerate(zip(slots, pins)):
        simd[w % 4] += W[s] + (PIN if p != 15 else 0)
    return max(simd), simd


cands = {"current": ([0, 1, 2, 10, 8, 9, 4, 3, 7, 5, 6], [0, 1, 2] + [15] * 8)}
# model search: ants on distinct SIMDs first (they start early), pins on light waves
best = []
light = [0, 1, 2, 5, 6, 7]
for ant_w in itertools.permutations(range(4), 3):
    for sp_w in itertools.permutations(range(4, 8), 2):
        free = [w for w in range(11) if w not in ant_w and w not in sp_w]
        slots = [None] * 11
        for w, s in zip(ant_w, (8, 9, 10)):
            slots[w] = s
        for w, s in zip(sp_w, (3, 4)):
            slots[w] = s
        for w, s in zip(free, light):
            slots[w] = s
        for pin_w in itertools.combinations(free, 3):
            pins = [15] * 11
            for i, w in enumerate(pin_w):
                pins[w] = i
            m, simd = load(slots, pins)
            best.append((m, slots[:], pins[:]))
best.sort(key=lambda t: t[0])
seen = set()
for m, slots, pins in best:
    key = tuple(sorted((w % 4, W[s], p != 15) for w, (s, p) in enumerate(zip(slots, pins))))
    if key in seen:
        continue
    seen.add(key)
    cands["m%02d_%.1f" % (len(cands), m)] = (slots, pins)
    if len(cands) >= 9:
        break
out = os.path.join(ROOT, "build", "abl")
os.makedirs(out, exist_ok=True)
for name, (slots, pins) in cands.items():
    print(name, slots, pins, "max SIMD load %.1f" % load(slots, pins)[0], load(slots, pins)[1], flush=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared",
                           "-DHIVE_WAVE_SLOTS=" + hexmap(slots), "-DHIVE_PIN_IDS=" + hexmap(pins),
                           "-o", os.path.join(out, "abl_%s.so" % name),
                           os.path.join(ROOT, "hive-alphazero_amd", "csrc", "hive_env.hip")])
