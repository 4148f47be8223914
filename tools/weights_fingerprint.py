#!/usr/bin/env python3
"""Fingerprints of "seed 0" synthetic weights on THIS host (round-3 diagnosis of BENCH_r02's self-check, see
profiles/r3_selfcheck_diagnosis.md): (a) torch.randn with a seeded CPU generator -- what whisper_compat.build_model used in
rounds 1-2; (b) nn.GRU / nn.Linear default initialisation from the unseeded global generator -- the rounds 1-2 bench head;
(c) the host-independent generator used since round 3.  Run it in the build container and on a GPU box and compare."""
import hashlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lyricalignment_amd import whisper_compat as wc


def sha(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


g = torch.Generator().manual_seed(0)
legacy = [0.02 * torch.randn((1024, 1024), generator=g) for _ in range(4)]
rnn = torch.nn.GRU(1024, 384, num_layers=2, bidirectional=True, batch_first=True)
fc = torch.nn.Linear(768, 21129)
wm = wc.build_model("tiny", seed=0)
hi = wc.HostIndependentRng(0)
print(json.dumps({"cpu": [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0],
                  "cpu_capability": torch.backends.cpu.get_cpu_capability(), "threads": torch.get_num_threads(),
                  "torch_randn_seed0_first4": legacy[0].flatten()[:4].tolist(), "torch_randn_seed0_sha": sha(legacy),
                  "default_init_head_sha": sha(list(rnn.parameters()) + list(fc.parameters())),
                  "default_init_fc_w0": fc.weight[0, :3].tolist(),
                  "host_independent_tiny_sha": sha(wm.state_dict().values()),
                  "host_independent_normal_sha": sha([hi.normal((1024, 1024))]), "host_independent_uniform_sha": sha([hi.uniform((999,))])}))
