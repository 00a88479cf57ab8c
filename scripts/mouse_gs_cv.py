"""`jx gs -BLUP -cv 5` on the reference's example data (example/mouse_hs1940, trait test0) -- the reference's README prints the
first fold of this command: "1    BLUP       0.704    0.675     0.493" (README.md:126-127).  Needs a GPU."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from janusx_amd import bed, cli  # noqa: E402


def main():
    d = np.load(os.path.join(ROOT, "tests", "golden", "mouse_hs1940.npz"))
    packed, ids = np.ascontiguousarray(d["packed"]), [str(s) for s in d["ids"]]
    m = packed.shape[0]
    with tempfile.TemporaryDirectory() as td:
        prefix = os.path.join(td, "mouse")
        bim = bed.Bim(["1"] * m, [f"s{j}" for j in range(m)], list(range(1, m + 1)), ["A"] * m, ["C"] * m)
        bed.write_bed(prefix, packed, ids, bim)
        with open(prefix + ".pheno", "w") as fh:
            fh.write("id\ttest0\n")
            for sid, v in zip(d["pheno_ids"], d["pheno"][:, 0]):
                fh.write(f"{sid}\t{'NA' if not np.isfinite(v) else repr(float(v))}\n")
        cli.main(["gs", "-bfile", prefix, "-p", prefix + ".pheno", "-BLUP", "-cv", "5", "-o", prefix])


main()
