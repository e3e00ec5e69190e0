"""tests/golden/well_far.json: WELL-1024a states beyond 2^32 words, produced by stepping the reference's own
well_1024a (oracle/well_far.c linked against oracle/_ref/libqvzref.so = /root/reference/src/well.c compiled in place).
Runs in the build container only (about two minutes of one core); the GPU box uses the committed vector.
BASELINE configs[4] (1B reads x 150 columns) draws 3.75e10 words: rank 7 of 8 starts at word 32 812 500 000."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
WORDS = [1000, (1 << 32) - 1, (1 << 32) + 7, (1 << 33) + 5, 875_000_000 * 150 // 4, (1 << 35) + 1, 37_499_990_000]


def main():
    exe = os.path.join(REF, "well_far")
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "oracle", "well_far.c"), "-L", REF, "-lqvzref",
                    "-Wl,-rpath," + REF], check=True)
    out = subprocess.run([exe] + [str(w) for w in WORDS], check=True, capture_output=True, text=True).stdout
    vec = json.loads(out)
    doc = dict(source="oracle/well_far.c stepping well_1024a of oracle/_ref/libqvzref.so (reference src/well.c:8-24) from 32 x 0x55555555",
               note="state/n as the reference holds them; rotated[i] = state[(n + i) & 31] is the n = 0 form the product stores",
               vectors=vec)
    json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "well_far.json"), "w"), indent=1)
    print("wrote", len(vec), "vectors")


if __name__ == "__main__":
    main()
