import sys, time, os, numpy as np
sys.path.insert(0, ".")
from janusx_amd import bed, cli
n, m = 5000, 50000
t0 = time.time()
packed, g = bed.synth_panel_numpy(n, m, seed=5, missing_rate=0.005)
y = bed.synth_phenotype(g, n_causal=50, pve=0.5, seed=5)
prefix = "/tmp/e2e"
ids = [f"id{i}" for i in range(n)]
bim = bed.Bim([str(1 + j * 20 // m) for j in range(m)], [f"rs{j}" for j in range(m)], list(range(1, m + 1)), ["C"] * m, ["T"] * m)
bed.write_bed(prefix, packed, ids, bim)
with open(prefix + ".pheno", "w") as fh:
    fh.write("id\ttraitA\n")
    for i in range(n):
        fh.write(f"{ids[i]}\t{float(y[i])!r}\n")
print(f"synth + write {time.time() - t0:.1f}s", flush=True)
for mode in ("-lmm", "-fvlmm", "-splmm-exact"):
    t0 = time.time()
    cli.main(["gwas", "-bfile", prefix, "-p", prefix + ".pheno", mode, "-o", prefix])
    print(f"CLI gwas {mode}: {time.time() - t0:.2f}s wall", flush=True)
t0 = time.time()
cli.main(["grm", "-bfile", prefix, "-o", prefix])
print(f"CLI grm: {time.time() - t0:.2f}s wall", flush=True)
t0 = time.time()
cli.main(["gs", "-bfile", prefix, "-p", prefix + ".pheno", "-BLUP", "-cv", "5", "-o", prefix])
print(f"CLI gs -BLUP -cv 5: {time.time() - t0:.2f}s wall", flush=True)
os.system("ls -la /tmp/e2e* | head -20; head -3 /tmp/e2e.traitA.lmm.tsv")
