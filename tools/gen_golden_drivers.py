#!/usr/bin/env python
"""Golden fixtures for the callers either side of the hot path, produced by EXECUTING the reference's
own statements in the build container.  The scripts cannot be imported (hydra / lightning / tqdm are absent), so the
statements are taken out of their AST in memory and run against stand-ins for what surrounds them; nothing of the
reference's text is stored — the fixture holds numbers and key names only.

  f2  scripts/generate_samples.py:117-152  the `match schedule_name:` statement (linear / cosine / edm / edm7) with the
      reference's `BSI` as `task.bsi`                       -> sched_<name>_<k> arrays, k in {8, 128}
  f1  scripts/eval_elbo.py:119-173         the `for steps in k:` loop (bpd bookkeeping, variance of the mean) and the
      `results = {...}` dictionary, fed with recorded per-sample (bpd, bpd_var) arrays -> means, variances, JSON text
  f3  bsi/tasks/ema_pytorch.py:196-201 + bsi/tasks/bsi.py:73-81   `create_ema(model)` around the reference DiT: the
      state-dict key list of a task-shaped module {model, ema_model} and `_extra_state` after n updates

Re-run:  python tools/gen_golden_drivers.py
"""
import ast
import importlib.util
import json
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ref = ref_shim.load_all()
REF = ref_shim.REF_ROOT


def find(tree, pred):
    return [n for n in ast.walk(tree) if pred(n)]


def run_nodes(nodes, env):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, "<reference statements>", "exec"), env)
    return env


class TinyConv(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.layer = torch.nn.Conv2d(4, 3, 3, padding=1)

    def forward(self, mu, t):
        t = torch.movedim(t.expand((1, *mu.shape[-2:], len(t))), -1, 0)
        return self.layer(torch.cat((mu, t), dim=-3))


def f2_schedules(out):
    src = open(os.path.join(REF, "scripts", "generate_samples.py")).read()
    tree = ast.parse(src)
    main = [f for f in find(tree, lambda n: isinstance(n, ast.FunctionDef))
            if find(f, lambda n: isinstance(n, ast.Match) and getattr(n.subject, "id", "") == "schedule_name")][0]
    body = main.body
    i_match = [i for i, n in enumerate(body) if isinstance(n, ast.Match)][0]
    i_if = i_match - 1
    assert isinstance(body[i_if], ast.If) and "max_variance" in ast.unparse(body[i_if])
    bsi = ref.BSI(TinyConv(), data_shape=(3, 8, 8), lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=8, preconditioning="edm",
                  discretization=ref.Discretization.image_8bit())
    for k in (8, 128):
        for name in ("linear", "cosine", "edm", "edm7"):
            env = {"task": types.SimpleNamespace(bsi=bsi), "schedule_name": name, "k": k, "device": torch.device("cpu"),
                   "torch": torch, "VDMTraining": type("VDMTraining", (), {}), "log": None}
            run_nodes([body[i_if], body[i_match]], env)
            out[f"sched_{name}_{k}"] = env["t"].detach().numpy()


class _Bar:
    """tqdm stand-in: iterable with the three methods the loop calls."""

    def __init__(self, it=None, **kw):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def set_description(self, *a, **k):
        pass

    def set_postfix(self, *a, **k):
        pass

    def close(self):
        pass


class _RecordedModel:
    """Replays recorded per-sample (bpd, bpd_var) arrays, one pair per call, as `elbo` / `finite_elbo` results."""

    def __init__(self, rec):
        self.rec = rec
        self.calls = []
        self.pos = defaultdict(int)

    def _next(self, key, x):
        bpd, var = self.rec[key][self.pos[key]]
        self.pos[key] += 1
        assert len(bpd) == len(x)
        return -torch.from_numpy(bpd), torch.from_numpy(bpd), {"bpd_var": torch.from_numpy(var)}

    def elbo(self, x, r, m, generator, estimate_var=False):
        self.calls.append(("inf", len(x), r, m, bool(estimate_var)))
        return self._next("inf", x)

    def finite_elbo(self, x, r, m, generator, estimate_var=False, t=None):
        self.calls.append((len(t) - 1, len(x), r, m, bool(estimate_var), [float(t[0]), float(t[-1])]))
        return self._next(len(t) - 1, x)


def f1_elbo_bookkeeping(out):
    src = open(os.path.join(REF, "scripts", "eval_elbo.py")).read()
    tree = ast.parse(src)
    loop = find(tree, lambda n: isinstance(n, ast.For) and getattr(n.target, "id", "") == "steps")[0]
    res = find(tree, lambda n: isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "results")[0]
    rng = np.random.default_rng(7)
    sizes = [5, 3, 4]
    ks = ["inf", 8, 32]
    rec = {k_: [((3.0 + 0.2 * rng.standard_normal(n)).astype(np.float32), (0.01 * rng.random(n)).astype(np.float32))
                for n in sizes] for k_ in ks}
    model = _RecordedModel(rec)
    dataloader = [(torch.zeros(n, 3, 4, 4), torch.zeros(n)) for n in sizes]
    env = {"k": ks, "tqdm": _Bar, "dataloader": dataloader, "move_data_to_device": lambda b, d: b, "device": torch.device("cpu"),
           "model": model, "r_samples": 2, "m_samples": 3, "generator": None, "VDM": type("VDM", (), {}), "torch": torch,
           "np": np, "bpd_means": defaultdict(lambda: np.zeros((0,))), "bpd_mean_vars": defaultdict(lambda: np.zeros((0,))),
           "k_bar": _Bar()}
    run_nodes([loop], env)
    env.update({"ckpt_path": "run/last.ckpt", "split": "test", "overrides": ["a=b"]})
    run_nodes([res], env)
    for k_ in ks:
        for i, (b, v) in enumerate(rec[k_]):
            out[f"elbo_rec_{k_}_{i}_bpd"], out[f"elbo_rec_{k_}_{i}_var"] = b, v
        out[f"elbo_mean_{k_}"] = np.float64(env["bpd_means"][k_])
        out[f"elbo_mean_var_{k_}"] = np.float64(env["bpd_mean_vars"][k_])
    out["elbo_results_json"] = np.array(json.dumps(env["results"]))
    out["elbo_calls_json"] = np.array(json.dumps(model.calls))
    out["elbo_batch_sizes"] = np.array(sizes)


def f3_checkpoint_keys(out):
    spec = importlib.util.spec_from_file_location("ref_ema", os.path.join(REF, "bsi", "tasks", "ema_pytorch.py"))
    ema_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ema_mod)
    # create_ema as bsi/tasks/bsi.py:73-81 writes it, taken from the file's AST (the module imports lightning / wandb)
    tsrc = open(os.path.join(REF, "bsi", "tasks", "bsi.py")).read()
    fn = find(ast.parse(tsrc), lambda n: isinstance(n, ast.FunctionDef) and n.name == "create_ema")[0]
    env = run_nodes([fn], {"EMA": ema_mod.EMA})
    torch.manual_seed(0)
    model = ref.dit.DenoisingDiT((3, 16, 16), 2, 128, 2, 2, dropout=None,
                                 fourier_features=ref.nn.FourierFeatures(n_min=6, n_max=8))
    ema = env["create_ema"](model, beta=0.9999, update_after_step=1000, update_every=1)
    task = torch.nn.Module()
    task.model = model
    task.ema_model = ema
    sd0 = task.state_dict()
    extra0 = dict(sd0["ema_model._extra_state"])
    for _ in range(5):
        ema.update()
    sd = task.state_dict()
    keys = list(sd.keys())
    out["ckpt_keys_json"] = np.array(json.dumps(keys))
    out["ckpt_extra_state_0_json"] = np.array(json.dumps(extra0))
    out["ckpt_extra_state_5_json"] = np.array(json.dumps(
        {k: (int(v) if not isinstance(v, bool) else v) for k, v in sd["ema_model._extra_state"].items()}))
    out["ckpt_shapes_json"] = np.array(json.dumps({k: list(v.shape) for k, v in sd.items() if hasattr(v, "shape")}))
    # one tensor pair to check prefixes address the right module: EMA copy equals online weights after the warm-up copies
    name = "dit.blocks.1.attn.to_qkv.weight"
    assert torch.equal(sd["model." + name], sd["ema_model.ema_model." + name])


if __name__ == "__main__":
    out = {}
    f2_schedules(out)
    f1_elbo_bookkeeping(out)
    f3_checkpoint_keys(out)
    out["_meta_torch_version"] = np.array(torch.__version__)
    path = os.path.join(OUT, "g12_drivers.npz")
    np.savez_compressed(path, **out)
    print(f"g12_drivers: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} entries")
