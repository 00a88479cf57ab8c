#!/usr/bin/env python3
"""One-off converter (build container only): /root/reference/example/mouse_hs1940.vcf.gz + .pheno
-> tests/golden/mouse_hs1940.npz (2-bit PLINK payload, BIM columns, sample ids, phenotype table).

BASELINE.json configs[0] / SURVEY.md §8(c): the real-data anchor for config C1. VCF parsing is out of scope for the
product (the reference does it in `jx gformat`); this script only maps GT 0/0, 0/1 (1/0), 1/1, ./. to the BED codes
00, 10, 11, 01 (dosage counts ALT = BIM column 6). The fixture holds DATA only."""
import gzip
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from janusx_amd import bed  # noqa: E402

SRC = "/root/reference/example/mouse_hs1940"


def main():
    chrom, pos, snp, a0, a1, rows = [], [], [], [], [], []
    gt_map = {"0/0": 0, "0|0": 0, "0/1": 1, "1/0": 1, "0|1": 1, "1|0": 1, "1/1": 2, "1|1": 2}
    with gzip.open(SRC + ".vcf.gz", "rt") as fh:
        for line in fh:
            if line.startswith("##"):
                continue
            f = line.rstrip("\n").split("\t")
            if line.startswith("#CHROM"):
                ids = f[9:]
                continue
            chrom.append(f[0]); pos.append(int(f[1])); snp.append(f[2]); a0.append(f[3]); a1.append(f[4])
            rows.append(np.array([gt_map.get(x.split(":")[0], -9) for x in f[9:]], dtype=np.int8))
    g = np.stack(rows)
    packed = bed.pack_dosage(g)
    with open(SRC + ".pheno") as fh:
        header = fh.readline().rstrip("\n").split("\t")[1:]
        pid, pv = [], []
        for line in fh:
            f = line.rstrip("\n").split("\t")
            pid.append(f[0])
            pv.append([float(x) if x not in ("NA", "") else np.nan for x in f[1:]])
    out = os.path.join(HERE, "mouse_hs1940.npz")
    np.savez_compressed(out, packed=packed, ids=np.array(ids), chrom=np.array(chrom), pos=np.array(pos),
                        snp=np.array(snp), a0=np.array(a0), a1=np.array(a1), pheno_ids=np.array(pid),
                        pheno_names=np.array(header), pheno=np.array(pv))
    print("wrote", out, os.path.getsize(out), "bytes; n =", len(ids), "m =", g.shape[0], "missing rate",
          float((g < 0).mean()))


if __name__ == "__main__":
    main()
