"""first-contact GPU check: every case in its own subprocess with a timeout; log to gpurun_out/first.log"""
import sys, os, gzip, glob, zlib, time, subprocess, struct, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LOG = os.path.join(ROOT, "gpurun_out", "first.log")

def child(path):
    import numpy as np
    from htslib_amd import _native as nat
    name, comp, exp = pickle.load(open(path, "rb"))
    eng = nat.Engine(0)
    try:
        t = time.time(); out, st = eng.bgzf_inflate_host(comp); dt = time.time() - t
        if out == exp: print("ok", name, len(comp), len(exp), "blocks", len(st), "%.1f ms" % (dt * 1e3)); return 0
        n = min(len(out), len(exp)); first = next((i for i in range(n) if out[i] != exp[i]), n)
        print("MISMATCH", name, len(out), len(exp), "first diff", first, out[first:first+16].hex(), exp[first:first+16].hex()); return 1
    except Exception as e:
        st = eng.last_status
        print("FAIL", name, e, "bad blocks:", np.flatnonzero(st != 0)[:10], st[st != 0][:10], "of", len(st)); return 1

def main():
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    log = open(LOG, "w")
    def P(*a):
        s = " ".join(str(x) for x in a); print(s, flush=True); log.write(s + "\n"); log.flush()
    P(subprocess.run("nproc; timeout 20 rocminfo | grep -E 'gfx|Compute Unit' | head -4; free -g | head -2", shell=True, capture_output=True, text=True).stdout)
    import numpy as np
    from htslib_amd import synth
    cases = []
    g = os.path.join(ROOT, "tests", "golden", "bgzf")
    for f in sorted(glob.glob(g + "/*.plain")):
        cases.append((os.path.basename(f), open(f[:-6], 'rb').read(), open(f, 'rb').read()))
    def blk(d, level=6, strategy=0):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        p = co.compress(d) + co.flush()
        return b"".join([synth._BGZF_HDR, struct.pack("<H", len(p) + 25), p, struct.pack("<II", zlib.crc32(d), len(d))])
    rng = np.random.default_rng(1)
    small = {"zeros": bytes(65280), "rand": rng.integers(0, 256, 60000, dtype=np.uint8).tobytes(),
             "text": (b"the quick brown fox jumps over the lazy dog\n" * 1500)[:65280], "one": b"x", "ab": b"ab" * 30000}
    for k, d in small.items():
        for strat in (0, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
            cases.append((f"{k} strat{strat}", blk(d, strategy=strat), d))
    data, bg = synth.bam_bgzf(4 << 20, threads=4)
    cases.append(("synth z6", bg, data))
    for lvl in (0, 1, 9):
        cases.append(("synth z%d" % lvl, synth.bgzf_compress(data[:1 << 20], level=lvl), data[:1 << 20]))
    open("/tmp/d.bin", "wb").write(data)
    for lvl in (1, 6, 9):
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_bgzip_ld"), "-l", str(lvl), "-c", "/tmp/d.bin"], capture_output=True)
        if r.returncode == 0: cases.append(("libdeflate l%d" % lvl, r.stdout, data))
        else: P("ref_bgzip_ld failed", r.stderr[:200])
    bad = 0
    for c in cases:
        pickle.dump(c, open("/tmp/case.pkl", "wb"))
        try:
            r = subprocess.run([sys.executable, __file__, "--child", "/tmp/case.pkl"], capture_output=True, text=True, timeout=25)
            P(r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode not in (0, 1) else "")
            bad += r.returncode != 0
        except subprocess.TimeoutExpired:
            P("HANG", c[0], len(c[1])); bad += 1
    P("BAD", bad)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child": sys.exit(child(sys.argv[2]))
    main()
