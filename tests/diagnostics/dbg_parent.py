"""Diagnostic: host builder vs device builder on a caterpillar walk; prints the first differences."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pagan2_msa_amd import host, synth
import oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from test_host_cpu import base_freq

names, seqs, nwk = synth.evolve_caterpillar(14, 150, seed=2)
bf = base_freq(seqs)
sq = dict(zip(names, seqs))
leaf_alpha, anc_alpha = host.alphabets(1)
count = [0]

def rec(t):
    if t[0] == "leaf":
        return host.HGraph.leaf(sq[t[1]], leaf_alpha), host.HGraph.leaf(sq[t[1]], leaf_alpha), (min(max(t[2], 0.001), 0.2) if t[2] > 0 else 0.001)
    hl, dl_, dl = rec(t[1])
    hr, dr_, dr = rec(t[2])
    model, pars = host.dna_model(bf, dl + dr)
    res = oracle.dp_align(hl.flatten(), hr.flatten(), model, None)
    hp = host.HGraph.parent(hl, hr, res, dl, dr, pars, 4, 0)
    dp = host.HGraph.parent_device(dl_, dr_, res, dl, dr, pars, 4, 0)
    sa, sd, ea, ef = hp.attrs(); sb, sdb, eb, efb = dp.attrs()
    print("node", count[0], "sites", sa.shape[0], "edges", ea.shape[0], eb.shape[0], "info", dp.build_info,
          "host nonreal", int((sa[:, 1] == 5).sum()))
    if sa.shape == sb.shape and not np.array_equal(sa, sb):
        bad = np.argwhere((sa != sb).any(axis=1))[:10, 0]
        for i in bad: print("  site", i, sa[i], sb[i])
    if ea.shape == eb.shape and not np.array_equal(ea, eb):
        bad = np.argwhere((ea != eb).any(axis=1))[:10, 0]
        for i in bad: print("  edge", i, ea[i], eb[i], ef[i], efb[i])
    count[0] += 1
    d = t[3]
    return hp, dp, (0.001 if d <= 0 else min(d, 0.2))

rec(synth.parse_newick(nwk))
