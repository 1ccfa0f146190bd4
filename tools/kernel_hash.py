"""Hash of the sources the curvature pass is built from: a PMC traffic figure measured in a separate rocprofv3 run
(tools/pmc_traffic.sh) is only quoted by bench.py while this hash still matches the sources in the tree."""
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'discrete-curvature-rewiring_amd', 'csrc')
PASS_SOURCES = ('dcr_bfc_common.h', 'dcr_internal.h', 'dcr_bfc.hip', 'dcr_bfc_nc.hip', 'dcr_bfc_h2.hip', 'dcr_graph.hip',
                'build.sh')


def pass_sources_hash(names=PASS_SOURCES):
    h = hashlib.sha256()
    for name in names:
        p = os.path.join(CSRC, name)
        if os.path.exists(p):
            h.update(name.encode())
            with open(p, 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(pass_sources_hash())
