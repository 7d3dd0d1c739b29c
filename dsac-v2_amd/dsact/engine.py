"""DsactEngine -- thin Python owner of one libdsact handle (one per GPU / process).

PyTorch-ROCm is used for exactly two things here: it OWNS the flat parameter / Adam / gradient
arenas (so `state_dict()`, `torch.save`, `torch.distributed.all_reduce` work on them unchanged), and
it provides the stream. All arithmetic of the update runs in the HIP kernels behind the C-ABI.
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import DsactError
from .layout import ArenaLayout, CnnArenaLayout

STAT_KEYS = [  # order of dsact_read_stats == dsac_v2.py:188-202
    "DSAC2/critic_avg_q1-RL iter", "DSAC2/critic_avg_q2-RL iter",
    "DSAC2/critic_avg_std1-RL iter", "DSAC2/critic_avg_std2-RL iter",
    "DSAC2/critic_avg_min_std1-RL iter", "DSAC2/critic_avg_min_std2-RL iter",
    "Loss/Actor loss-RL iter", "Loss/Critic loss-RL iter",
    "DSAC2/policy_mean-RL iter", "DSAC2/policy_std-RL iter", "DSAC2/entropy-RL iter",
    "DSAC2/alpha-RL iter", "DSAC2/mean_std1", "DSAC2/mean_std2",
]


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class DsactEngine:
    def __init__(self, obs_dim: int, act_dim: int, hidden: Sequence[int], batch: int, *,
                 gamma=0.99, tau=0.005, tau_b=None, auto_alpha=True, alpha=0.2, delay_update=2,
                 lr_q=1e-4, lr_pi=1e-4, lr_alpha=3e-4, min_log_std=-20.0, max_log_std=0.5,
                 global_batch: Optional[int] = None, device: int = 0, conv_type: Optional[str] = None,
                 algo: str = "DSAC_V2", td_bound: float = 20.0, v1_bound: bool = True, value_act: int = 0, policy_act: int = 0, act_dist: int = 0,
                 policy_std_type: str = "mlp_shared", value_out_act: int = 0, policy_out_act: int = 0,
                 policy_hidden: Optional[Sequence[int]] = None, pad_widths: bool = False):
        """obs_dim: int for the MLP nets; with `conv_type` ("type_1" / "type_2", reference
        networks/cnn.py:173-228) the (C, H, W) image shape, and `hidden` must be that type's MLP widths.

        pad_widths (round 6): hidden widths the row-slice chain kernels do not take as they are -- ragged ones (96, 40), unequal
        value / policy lists of the same depth -- are STORED zero-padded to one common width of 64 / 128 / 256 when that puts
        the update on the chains (ArenaLayout pad_to: the reference's tensors are windows of the stored ones, the padding stays an
        exact zero). Falls back to the exact layout (tile-stage kernels) when the chains refuse the padded shape anyway."""
        kw = dict(gamma=gamma, tau=tau, tau_b=tau_b, auto_alpha=auto_alpha, alpha=alpha, delay_update=delay_update, lr_q=lr_q, lr_pi=lr_pi,
                  lr_alpha=lr_alpha, min_log_std=min_log_std, max_log_std=max_log_std, global_batch=global_batch, device=device,
                  conv_type=conv_type, algo=algo, td_bound=td_bound, v1_bound=v1_bound, value_act=value_act, policy_act=policy_act,
                  act_dist=act_dist, policy_std_type=policy_std_type, value_out_act=value_out_act, policy_out_act=policy_out_act,
                  policy_hidden=policy_hidden)
        pad = self._pad_width(hidden, policy_hidden, batch, obs_dim, **{k: v for k, v in kw.items() if k != "policy_hidden"}) if pad_widths else None
        self._init(obs_dim, act_dim, hidden, batch, pad_to=pad, **kw)
        if pad and not self.chain_active:     # the chains refused it (LDS, an environment switch, ...): no reason to carry the padding
            self.close()
            self._init(obs_dim, act_dim, hidden, batch, pad_to=None, **kw)

    @staticmethod
    def _pad_width(hidden, policy_hidden, batch, obs_dim, *, conv_type=None, algo="DSAC_V2", policy_std_type="mlp_shared", value_act=0,
                   policy_act=0, **_):
        """the common stored width that puts this configuration on the row-slice chains, or None (no padding needed / possible):
        the shape conditions of dsact_create's chain_ok (csrc/dsact_api.hip) + act(0) == 0 for both hidden activations"""
        ph = list(policy_hidden) if policy_hidden is not None else list(hidden)
        widths = list(hidden) + ph
        if conv_type or policy_std_type == "mlp_separated" or len(ph) != len(hidden) or len(hidden) > 4:
            return None
        if len(set(widths)) == 1 and widths[0] in (64, 128, 256):
            return None                                   # the chains take it as it is
        if max(widths) > 256 or 4 in (int(value_act), int(policy_act)):   # 4 = sigmoid: act(0) = 0.5 would make the padding live
            return None
        if batch % 16 or (batch > 256 and batch % 256):
            return None
        return 64 if max(widths) <= 64 else (128 if max(widths) <= 128 else 256)

    def _init(self, obs_dim, act_dim, hidden, batch, *, pad_to=None,
              gamma=0.99, tau=0.005, tau_b=None, auto_alpha=True, alpha=0.2, delay_update=2,
              lr_q=1e-4, lr_pi=1e-4, lr_alpha=3e-4, min_log_std=-20.0, max_log_std=0.5,
              global_batch=None, device=0, conv_type=None,
              algo="DSAC_V2", td_bound=20.0, v1_bound=True, value_act=0, policy_act=0, act_dist=0,
              policy_std_type="mlp_shared", value_out_act=0, policy_out_act=0, policy_hidden=None):
        import torch

        self._lib = _ffi.load()
        if not torch.cuda.is_available():
            raise DsactError("DsactEngine needs an MI355X (torch.cuda.is_available() is False); "
                             "the DSAC-T update has no CPU fallback")
        self.torch = torch
        self.conv_type = conv_type
        self.algo = algo
        if algo not in ("DSAC_V2", "DSAC_V1"):
            raise DsactError("algo must be DSAC_V2 or DSAC_V1")
        if conv_type and policy_std_type != "mlp_shared":
            raise DsactError("policy_std_type %r is built for the MLP nets only" % policy_std_type)
        if conv_type:
            self.layout = CnnArenaLayout(obs_dim, act_dim, conv_type, n_critics=2 if algo == "DSAC_V2" else 1)
            if list(hidden) != self.layout.hidden:
                raise DsactError("conv_type %s fixes the MLP widths to %s" % (conv_type, self.layout.hidden))
            self.obs_shape = self.layout.obs_shape
            obs_dim = self.layout.obs_dim
        else:
            self.layout = ArenaLayout(obs_dim, act_dim, list(hidden), n_critics=2 if algo == "DSAC_V2" else 1,
                                      policy_std_type=policy_std_type, policy_hidden=policy_hidden, pad_to=pad_to)
            self.obs_shape = (int(obs_dim),)
        self.obs_dim, self.act_dim, self.batch = int(obs_dim), int(act_dim), int(batch)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        cfg = _ffi.Config()
        cfg.obs_dim, cfg.act_dim, cfg.n_hidden = obs_dim, act_dim, len(hidden)
        if len(hidden) > _ffi.MAX_HIDDEN:
            raise DsactError("at most %d hidden layers" % _ffi.MAX_HIDDEN)
        for i, w in enumerate(hidden):
            cfg.hidden[i] = int(pad_to or w)      # (pad_to: the library sees the STORED widths -- an equal-width configuration)
        cfg.batch = batch
        cfg.global_batch = int(global_batch or batch)
        cfg.auto_alpha = 1 if auto_alpha else 0
        cfg.delay_update = int(delay_update)
        cfg.gamma, cfg.tau = gamma, tau
        cfg.tau_b = tau if tau_b is None else tau_b
        cfg.lr_q, cfg.lr_pi, cfg.lr_alpha = lr_q, lr_pi, lr_alpha
        cfg.alpha_fixed = alpha
        cfg.min_log_std, cfg.max_log_std = min_log_std, max_log_std
        cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps = 0.9, 0.999, 1e-8
        if conv_type:
            cfg.conv_type = self.layout.conv_id
            cfg.img_c, cfg.img_h, cfg.img_w = self.obs_shape
        cfg.algo = 0 if algo == "DSAC_V2" else 1
        cfg.td_bound = float(td_bound)
        cfg.v1_unbounded = 0 if v1_bound else 1
        cfg.value_act, cfg.policy_act = int(value_act), int(policy_act)   # hidden activations: 0 gelu .. 5 tanh (include/dsact.h)
        cfg.policy_std_param = 1 if policy_std_type == "parameter" else 0   # networks/mlp.py:63-73 (include/dsact.h)
        cfg.policy_twin = 1 if policy_std_type == "mlp_separated" else 0    # networks/mlp.py:46-57: two MLPs side by side
        if policy_hidden is not None and list(policy_hidden) != list(hidden) and not pad_to:   # value_hidden_sizes != policy_hidden_sizes (include/dsact.h)
            if conv_type or not 1 <= len(policy_hidden) <= _ffi.MAX_HIDDEN:
                raise DsactError("policy_hidden needs the MLP nets and 1..%d layers" % _ffi.MAX_HIDDEN)
            for i, w in enumerate(policy_hidden):
                cfg.policy_hidden[i] = int(w)
            if len(policy_hidden) != len(hidden):    # lists of different depth (include/dsact.h policy_n_hidden)
                cfg.policy_n_hidden = len(policy_hidden)
        cfg.value_out_act, cfg.policy_out_act = int(value_out_act), int(policy_out_act)   # 0 linear, 1..5 relu .. tanh (include/dsact.h)
        cfg.act_dist = int(act_dist)                                       # 0 TanhGaussDistribution, 1 GaussDistribution
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = self._lib.dsact_create(C.byref(cfg), self.device_index, C.byref(self._h))
        if rc != 0:
            msg = self._lib.dsact_last_error(self._h).decode() if self._h else ""
            if self._h:
                self._lib.dsact_destroy(self._h)
                self._h = C.c_void_p()
            raise DsactError("dsact_create failed (%s): %s" % (_ffi.E_NAMES.get(rc, rc), msg))
        lay = self.layout
        assert self._lib.dsact_online_count(self._h) == lay.n_online, (self._lib.dsact_online_count(self._h), lay.n_online)
        assert self._lib.dsact_target_count(self._h) == lay.n_target
        assert self._lib.dsact_q_count(self._h) == lay.n_q and self._lib.dsact_pi_count(self._h) == lay.n_pi
        # arenas: torch tensors, device pointers handed to the library
        dev = self.device
        self.online = torch.zeros(lay.n_online, dtype=torch.float32, device=dev)
        self.target = torch.zeros(lay.n_target, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(lay.n_online, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(lay.n_online, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(lay.n_online + 2, dtype=torch.float32, device=dev)
        self._chk(self._lib.dsact_bind_arenas(
            self._h, self.online.data_ptr(), self.target.data_ptr(), self.adam_m.data_ptr(),
            self.adam_v.data_ptr(), self.grads.data_ptr()))
        self._graph_steps = 0
        self.stage_serial = 0
        self.buffer_capacity = 0
        self.rows_added = 0      # rows ever written to the ring (HipBatch tokens detect overwritten rows with it)
        self.fill_epoch = 0      # bumped by buffer_fill_device (writes at an arbitrary row: outstanding tokens become invalid)

    # ---- plumbing -----------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise DsactError("%s: %s" % (_ffi.E_NAMES.get(rc, rc), self._lib.dsact_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dsact_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_torch_stream(self):
        """Run on torch's current stream. The legacy default stream has no handle to pass (its `cuda_stream` is 0,
        which `dsact_set_stream` reads as "handle-owned stream"): enter a `torch.cuda.stream(...)` context first, or --
        simpler -- leave the engine on its own stream and issue the torch work on `engine.torch_stream`."""
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        if not s:
            raise DsactError("torch's current stream is the legacy default stream (handle 0); run torch work under "
                             "`with torch.cuda.stream(engine.torch_stream):` instead")
        self._chk(self._lib.dsact_set_stream(self._h, C.c_void_p(s)))
        self._torch_stream = None

    @property
    def stream_ptr(self) -> int:
        return int(self._lib.dsact_get_stream(self._h) or 0)

    @property
    def torch_stream(self):
        """The engine's HIP stream as a torch stream: torch ops issued under `torch.cuda.stream(engine.torch_stream)`
        (collectives on `engine.grads`, tensor math on the arenas) are stream-ordered with the engine's kernels."""
        ts = getattr(self, "_torch_stream", None)
        if ts is None or ts.cuda_stream != self.stream_ptr:
            ts = self.torch.cuda.ExternalStream(self.stream_ptr, device=self.device)
            self._torch_stream = ts
        return ts

    def sync(self):
        self._chk(self._lib.dsact_sync(self._h))

    def set_action_limits(self, high, low):
        hi, lo = _f32(high), _f32(low)
        self._chk(self._lib.dsact_set_action_limits(self._h, _ffi.fptr(hi), _ffi.fptr(lo)))

    # ---- state ------------------------------------------------------------------------------------
    def get_state(self):
        steps = (C.c_int32 * 3)()
        ms = (C.c_float * 2)()
        self._chk(self._lib.dsact_get_state(self._h, steps, ms))
        return {"adam_steps": [int(s) for s in steps], "mean_std": [float(ms[0]), float(ms[1])]}

    def set_state(self, adam_steps=None, mean_std=None):
        st = (C.c_int32 * 3)(*adam_steps) if adam_steps is not None else None
        ms = (C.c_float * 2)(*mean_std) if mean_std is not None else None
        self._chk(self._lib.dsact_set_state(self._h, st, ms))

    HYPER = {"gamma": 0, "tau": 1, "tau_b": 2, "auto_alpha": 3, "alpha": 4, "delay_update": 5, "TD_bound": 6, "bound": 7}

    def set_hyper(self, name: str, value):
        """dsact_set_hyper: the next enqueued update uses `value`; a captured graph is dropped (build it again)"""
        self._chk(self._lib.dsact_set_hyper(self._h, self.HYPER[name], float(value)))
        self._graph_steps = 0

    # ---- replay ring -------------------------------------------------------------------------------
    def buffer_create(self, capacity: int):
        self._chk(self._lib.dsact_buffer_create(self._h, int(capacity)))
        self.buffer_capacity = int(capacity)

    def buffer_add(self, obs, act, rew, obs2, done, logp=None):
        obs, act, rew, obs2, done = _f32(obs), _f32(act), _f32(rew), _f32(obs2), _f32(done)
        n = int(rew.shape[0])
        lp = _f32(logp) if logp is not None else None
        self._chk(self._lib.dsact_buffer_add(self._h, n, _ffi.fptr(obs), _ffi.fptr(act), _ffi.fptr(rew),
                                             _ffi.fptr(obs2), _ffi.fptr(done), _ffi.fptr(lp)))
        self.rows_added += n

    def buffer_fill_device(self, row0, obs, act, rew, obs2, done):
        """rows from torch CUDA tensors (synthetic benchmark buffers stay on the device)."""
        n = int(rew.shape[0])
        for t in (obs, act, rew, obs2, done):
            assert t.is_cuda and t.dtype == self.torch.float32 and t.is_contiguous()
        self.torch.cuda.current_stream(self.device).synchronize()
        self._chk(self._lib.dsact_buffer_fill_device(self._h, int(row0), n, obs.data_ptr(), act.data_ptr(),
                                                     rew.data_ptr(), obs2.data_ptr(), done.data_ptr()))
        self.rows_added += n
        self.fill_epoch += 1

    @property
    def buffer_size(self):
        return int(self._lib.dsact_buffer_size(self._h))

    @property
    def buffer_ptr(self):
        return int(self._lib.dsact_buffer_ptr(self._h))

    def gather(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64))
        self._chk(self._lib.dsact_gather(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), int(idx.shape[0])))
        self.stage_serial += 1   # which minibatch sits in the staging set (HipBatch tokens compare against it)

    def read_batch(self, with_logp=True) -> Dict[str, np.ndarray]:
        B, O, A = self.batch, self.obs_dim, self.act_dim
        shp = (B,) + tuple(self.obs_shape)
        out = {"obs": np.empty(shp, np.float32), "act": np.empty((B, A), np.float32),
               "rew": np.empty(B, np.float32), "obs2": np.empty(shp, np.float32),
               "done": np.empty(B, np.float32)}
        lp = np.empty(B, np.float32) if with_logp else None
        self._chk(self._lib.dsact_read_batch(self._h, _ffi.fptr(out["obs"]), _ffi.fptr(out["act"]),
                                             _ffi.fptr(out["rew"]), _ffi.fptr(out["obs2"]),
                                             _ffi.fptr(out["done"]), _ffi.fptr(lp)))
        if with_logp:
            out["logp"] = lp
        return out

    def _src(self, a, n):
        """float* for dsact_load_batch: a CUDA tensor on this engine's GPU is passed by device address (no trip
        through the host), anything else as a contiguous host float32 array. Returns (pointer, keep-alive)."""
        torch = self.torch
        if isinstance(a, torch.Tensor) and a.is_cuda and a.device.index == self.device_index:
            t = a.detach().to(torch.float32).contiguous()
            assert t.numel() == n, (tuple(t.shape), n)
            return C.cast(C.c_void_p(t.data_ptr()), _ffi._FP), t
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        a = _f32(a)
        assert a.size == n, (a.shape, n)
        return _ffi.fptr(a), a

    def load_batch(self, obs, act, rew, obs2, done):
        """Stage a minibatch that lives outside the HIP ring: numpy / CPU tensors or CUDA tensors (mixed is fine)."""
        torch, B = self.torch, self.batch
        srcs = [self._src(obs, B * self.obs_dim), self._src(act, B * self.act_dim), self._src(rew, B),
                self._src(obs2, B * self.obs_dim), self._src(done, B)]
        if any(isinstance(k, torch.Tensor) for _, k in srcs):
            torch.cuda.current_stream(self.device).synchronize()   # the producers of the CUDA sources have finished
        self._chk(self._lib.dsact_load_batch(self._h, *[p for p, _ in srcs]))
        self.stage_serial += 1

    def upload_index_table(self, idx):
        idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64))
        assert idx.ndim == 2 and idx.shape[1] == self.batch
        self._chk(self._lib.dsact_upload_index_table(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     int(idx.shape[0])))

    # ---- noise ---------------------------------------------------------------------------------------
    def set_noise(self, eps_new, eps_2, z5, z6):
        a, b, c, d = _f32(eps_new), _f32(eps_2), _f32(z5), _f32(z6)
        assert a.shape == (self.batch, self.act_dim) and c.shape == (self.batch,)
        self._chk(self._lib.dsact_set_noise(self._h, _ffi.fptr(a), _ffi.fptr(b), _ffi.fptr(c), _ffi.fptr(d)))

    def set_device_rng(self, seed: int):
        self._chk(self._lib.dsact_set_device_rng(self._h, int(seed)))

    # ---- update ---------------------------------------------------------------------------------------
    def compute_grads(self, iteration: int, flags: int = 0):
        self._chk(self._lib.dsact_compute_grads(self._h, int(iteration), int(flags)))

    def apply_update(self, iteration: int):
        self._chk(self._lib.dsact_apply_update(self._h, int(iteration)))

    def step(self, iteration: int, flags: int = 0):
        self._chk(self._lib.dsact_step(self._h, int(iteration), int(flags)))

    def graph_build(self, steps_per_graph: int = 2, flags: int = 0):
        self._chk(self._lib.dsact_graph_build(self._h, int(steps_per_graph), int(flags)))
        self._graph_steps = int(steps_per_graph)

    def graph_run(self, first_iteration: int, n_steps: int):
        self.stage_serial += 1   # the index table's device-side gather replaces the staged minibatch
        self._chk(self._lib.dsact_graph_run(self._h, int(first_iteration), int(n_steps)))

    def run_group(self, first_iteration: int, idx_rows, noise_rows=None, flags: int = 0):
        """dsact_run_group: len(idx_rows) x { sample_batch -> local_update } of the reference's loop between two sampler calls
        as one graph replay. idx_rows int64 [n][batch] (drawn by the caller with the reference's np.random.randint calls);
        noise_rows float32 [n][2*B*A + 2*B] (eps_new | eps_2 | z5 | z6 per update) or None for device Philox noise.
        Asynchronous."""
        idx = np.ascontiguousarray(np.asarray(idx_rows, dtype=np.int64))
        assert idx.ndim == 2 and idx.shape[1] == self.batch, idx.shape
        n = int(idx.shape[0])
        nz = None
        if noise_rows is not None:
            nz = _f32(noise_rows)
            assert nz.shape == (n, 2 * self.batch * self.act_dim + 2 * self.batch), nz.shape
        self.stage_serial += 1   # the group's last minibatch replaces the staged one
        self._chk(self._lib.dsact_run_group(self._h, int(first_iteration), n, idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                            _ffi.fptr(nz), int(flags)))
        self._graph_steps = n

    # data-parallel halves (iteration and index-table row come from device state)
    def dp_begin(self, first_iteration: int):
        self._chk(self._lib.dsact_dp_begin(self._h, int(first_iteration)))

    def dp_grads(self, flags: int = 0):
        self.stage_serial += 1   # the index table's device-side gather replaces the staged minibatch
        self._chk(self._lib.dsact_dp_enqueue_grads(self._h, int(flags)))

    def dp_grads_critic(self, flags: int = 0):
        """first half of dp_grads: afterwards grads[:2*n_q] (q1 | q2) is final"""
        self.stage_serial += 1   # the index table's device-side gather replaces the staged minibatch
        self._chk(self._lib.dsact_dp_enqueue_grads_critic(self._h, int(flags)))

    def dp_grads_actor(self, flags: int = 0):
        self._chk(self._lib.dsact_dp_enqueue_grads_actor(self._h, int(flags)))

    @property
    def critic_grad_count(self):
        return 2 * self.layout.n_q

    def dp_set_strict(self, enable: bool = True):
        """strict data-parallel mode: `std_sums` (2 floats, torch-owned) is all-reduced by the caller between
        dp_forward() and dp_backward()"""
        if enable:
            self.std_sums = self.torch.zeros(2, dtype=self.torch.float32, device=self.device)
            self._chk(self._lib.dsact_dp_set_strict(self._h, self.std_sums.data_ptr()))
        else:
            self._chk(self._lib.dsact_dp_set_strict(self._h, None))
            self.std_sums = None

    def dp_forward(self, flags: int = 0):
        self.stage_serial += 1   # the index table's device-side gather replaces the staged minibatch
        self._chk(self._lib.dsact_dp_enqueue_forward(self._h, int(flags)))

    def dp_backward(self, flags: int = 0):
        self._chk(self._lib.dsact_dp_enqueue_backward(self._h, int(flags)))

    def dp_apply(self):
        self._chk(self._lib.dsact_dp_enqueue_apply(self._h))

    # native collective: RCCL opened by the library itself (the copy torch ships, so one RCCL serves the process)
    @staticmethod
    def _rccl_path():
        import torch

        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return p.encode() if os.path.exists(p) else None

    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = self._lib.dsact_comm_unique_id(self._rccl_path(), buf)
        if rc != 0:
            raise DsactError("dsact_comm_unique_id failed (%s)" % _ffi.E_NAMES.get(rc, rc))
        return bytes(buf)

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128)(*unique_id)
        self._chk(self._lib.dsact_comm_init(self._h, int(rank), int(world), buf, self._rccl_path()))
        self.comm_world = int(world)

    def comm_destroy(self):
        self._chk(self._lib.dsact_comm_destroy(self._h))

    def dp_allreduce(self):
        """average of the gradient arena (+ mean_std tail) over the ranks, on the engine's stream (RCCL)"""
        self._chk(self._lib.dsact_dp_enqueue_allreduce(self._h))

    STATS_SLOTS = 16

    def stats_snapshot(self, slot: int):
        """asynchronous: reduce the last update's statistics into ring slot `slot` (read later with read_stats(slot))"""
        self._chk(self._lib.dsact_stats_snapshot(self._h, int(slot) % self.STATS_SLOTS))

    def read_stats(self, slot=None) -> Dict[str, float]:
        out = (C.c_float * 16)()
        if slot is None:
            self._chk(self._lib.dsact_read_stats(self._h, out))
        else:
            self._chk(self._lib.dsact_stats_read(self._h, int(slot) % self.STATS_SLOTS, out))
        d = {k: float(out[i]) for i, k in enumerate(STAT_KEYS)}
        d["_iteration"] = float(out[14])
        d["_device_ms"] = float(out[15]) if slot is None else -1.0   # stream time of the last eager update (-1: unknown)
        return d

    # ---- measurement ------------------------------------------------------------------------------------
    def time_steps(self, first_iteration: int, n_steps: int, use_graph: bool = True, flags: int = 0) -> float:
        ms = C.c_float()
        self.stage_serial += 1   # the index table's device-side gather replaces the staged minibatch
        self._chk(self._lib.dsact_time_steps(self._h, int(first_iteration), int(n_steps), int(flags),
                                             1 if use_graph else 0, C.byref(ms)))
        return float(ms.value)

    def time_stage(self, stage: int, reps: int = 200):
        """(milliseconds for `reps` launches, multiply-accumulates per launch) of one forward tile stage"""
        ms, macs = C.c_float(), C.c_double()
        self._chk(self._lib.dsact_time_stage(self._h, int(stage), int(reps), C.byref(ms), C.byref(macs)))
        return float(ms.value), float(macs.value)

    @property
    def chain_active(self) -> bool:
        """True when this engine runs the update as row-slice fused chains (csrc/dsact_chain.h)"""
        return bool(self._lib.dsact_chain_active(self._h))

    def profile_step(self, iteration: int, flags: int = 0):
        arr = (_ffi.KernelTime * 128)()
        n = C.c_int32()
        self.stage_serial += 1   # the index table's device-side gather replaces the staged minibatch
        self._chk(self._lib.dsact_profile_step(self._h, int(iteration), int(flags), arr, 128, C.byref(n)))
        return [(arr[i].name.decode(), float(arr[i].ms), int(arr[i].blocks)) for i in range(n.value)]

    def profile_steps(self, first_iteration: int, n_steps: int, flags: int = 0):
        """per-dispatch (name, ms, blocks) of n_steps updates issued eagerly as the sequence graph_build(n_steps) would
        capture (the pipelined one where eligible)"""
        arr = (_ffi.KernelTime * 512)()
        n = C.c_int32()
        self.stage_serial += 1
        self._chk(self._lib.dsact_profile_steps(self._h, int(first_iteration), int(n_steps), int(flags), arr, 512, C.byref(n)))
        return [(arr[i].name.decode(), float(arr[i].ms), int(arr[i].blocks)) for i in range(n.value)]

    def debug_read(self, name: str, cap: int = 1 << 24) -> np.ndarray:
        buf = np.empty(cap, np.float32)
        n = C.c_size_t()
        self._chk(self._lib.dsact_debug_read(self._h, name.encode(), _ffi.fptr(buf), cap, C.byref(n)))
        return buf[: n.value].copy()

    def debug_set(self, name: str, value: float):
        self._chk(self._lib.dsact_debug_set(self._h, name.encode(), float(value)))
        if name in ("withhold_flag", "fwd_merge"):
            self._graph_steps = 0   # the library dropped the captured graph

    def debug_get(self, name: str) -> float:
        v = C.c_double()
        self._chk(self._lib.dsact_debug_get(self._h, name.encode(), C.byref(v)))
        return float(v.value)

    def policy_dirty(self):
        """the caller wrote policy parameters with torch ops on the arena (load_state_dict, a manual copy_): the host-side
        acting snapshot (csrc/dsact_host_act.h) is refreshed before the next acting call"""
        self._chk(self._lib.dsact_debug_set(self._h, b"policy_dirty", 1.0))

    def note_torch_writes(self, params):
        """torch bumps a tensor's version counter on every in-place write: a changed sum over the policy's parameters
        since the last look means somebody wrote them outside the library"""
        v = sum(p._version for p in params)
        if v != getattr(self, "_param_versions", None):
            self._param_versions = v
            self.policy_dirty()

    def act_sample(self, obs, eps):
        """dsact_act_sample: (action float32[A], logp float) of TanhGaussDistribution.sample() for ONE observation with the
        caller's N(0,1) draw eps[A]; obs / eps are contiguous float32 arrays (no copies are made here)"""
        if getattr(self, "_act_out", None) is None:
            self._act_out = np.empty(self.act_dim, np.float32)
            self._act_lp = np.empty(1, np.float32)
            self._act_out_p, self._act_lp_p = _ffi.fptr(self._act_out), _ffi.fptr(self._act_lp)
        rc = self._lib.dsact_act_sample(self._h, obs.ctypes.data_as(_ffi._FP), eps.ctypes.data_as(_ffi._FP),
                                        self._act_out_p, self._act_lp_p)
        if rc != 0:
            self._chk(rc)
        return self._act_out, self._act_lp

    def act_sample_addr(self, obs_addr: int, eps_addr: int, act_addr: int, logp_addr: int):
        """dsact_act_sample on plain integer addresses (the sampler's per-step call: no array or pointer objects are made;
        results land in the caller's rows). A second binding of the same symbol whose arguments are void*."""
        f = getattr(self, "_act_addr_fn", None)
        if f is None:
            f = self._lib["dsact_act_sample"]          # a fresh function object: its argtypes are its own
            f.restype, f.argtypes = C.c_int, [C.c_void_p] * 5
            self._act_addr_fn = f
        rc = f(self._h, obs_addr, eps_addr, act_addr, logp_addr)
        if rc != 0:
            self._chk(rc)

    def policy_forward(self, obs) -> np.ndarray:
        obs = _f32(obs).reshape(-1, self.obs_dim)
        n = obs.shape[0]
        out = np.empty((n, 2 * self.act_dim), np.float32)
        for s in range(0, n, 64):
            e = min(n, s + 64)
            chunk = np.ascontiguousarray(obs[s:e])
            o = np.empty((e - s, 2 * self.act_dim), np.float32)
            self._chk(self._lib.dsact_policy_forward(self._h, _ffi.fptr(chunk), e - s, _ffi.fptr(o)))
            out[s:e] = o
        return out


# process-wide registry: the replay buffer plugin shares the algorithm's handle so that a gathered
# minibatch never leaves HBM (mix-and-match case iii)
_CURRENT = None


def register_engine(engine: "DsactEngine"):
    global _CURRENT
    _CURRENT = engine


def current_engine() -> Optional["DsactEngine"]:
    return _CURRENT
