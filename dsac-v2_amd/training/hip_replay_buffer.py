"""HipReplayBuffer -- device-resident replay ring; drop-in for reference training/replay_buffer.py.

Discovered by the reference's rule `create_buffer(buffer_name="hip_replay_buffer")` -> module
`training.hip_replay_buffer`, class `HipReplayBuffer` (reference utils/initialization.py:90-117).

Same observable behaviour as the reference ReplayBuffer (replay_buffer.py:20-90): SoA fp32 ring with
`ptr=(ptr+1)%N`, `size=min(size+1,N)`, uniform index draw with replacement from the GLOBAL legacy
NumPy RandomState (`np.random.randint(0, size, batch)`, bit-exact by construction because the very
same call is made here on the host), `sample_batch` returns a dict with keys
obs/obs2/act/rew/done/logp. The rows live in HBM; the gather is a HIP kernel (k_gather) that writes
straight into the update's staging area, and the returned `HipBatch` is a token for it.
"""
import numpy as np

from dsact.engine import DsactEngine, current_engine

__all__ = ["HipReplayBuffer"]


class HipReplayBuffer:
    def __init__(self, index=0, **kwargs):
        self.obsv_dim = kwargs["obsv_dim"]   # int, or the (C, H, W) tuple of the CNN configs (replay_buffer.py:22-31)
        self._obs_shape = tuple(self.obsv_dim) if isinstance(self.obsv_dim, (tuple, list)) else (int(self.obsv_dim),)
        self._obs_flat = int(np.prod(self._obs_shape))
        self.act_dim = kwargs["action_dim"]
        self.max_size = int(kwargs["buffer_max_size"])
        if kwargs.get("additional_info"):
            raise NotImplementedError("additional_info is not supported by HipReplayBuffer")
        eng = kwargs.get("hip_engine") or current_engine()
        B = int(kwargs["replay_batch_size"])
        if eng is None or eng.obs_dim != self._obs_flat or eng.act_dim != self.act_dim or eng.batch != B:
            if len(self._obs_shape) == 3:
                from dsact.layout import CONV_TYPES
                ct = kwargs.get("value_conv_type", "type_2")
                eng = DsactEngine(self._obs_shape, self.act_dim, CONV_TYPES[ct][4], B, conv_type=ct,
                                  device=int(kwargs.get("hip_device", 0)))
            else:
                hidden = list(kwargs.get("value_hidden_sizes", [32]))
                eng = DsactEngine(self._obs_flat, self.act_dim, hidden, B, device=int(kwargs.get("hip_device", 0)))
        self.engine = eng
        self.engine.buffer_create(self.max_size)

    @property
    def size(self):
        return self.engine.buffer_size

    @property
    def ptr(self):
        return self.engine.buffer_ptr

    def __len__(self):
        return self.size

    def __get_RAM__(self):
        row_bytes = 4 * (2 * self._obs_flat + self.act_dim + 3)
        return row_bytes * self.size / 1e6  # MB resident in HBM

    def store(self, obs, info, act, rew, next_obs, done, logp, next_info):
        self.add_batch([(obs, info, act, rew, next_obs, done, logp, next_info)])

    def add_batch(self, samples: list):
        n = len(samples)
        if n == 0:
            return
        packed = getattr(samples, "packed", None)
        if packed is not None and packed[0].shape == (n, self._obs_flat) and self._packed_matches(samples, packed):
            # HipOffSampler's fast path already holds the transitions as packed float32 arrays (training/hip_sampler.py)
            obs, act, rew, obs2, done, logp = packed
            self.engine.buffer_add(obs, act, rew, obs2, done, logp)
            return
        O, A = self._obs_flat, self.act_dim
        obs = np.empty((n, O), np.float32)
        obs2 = np.empty((n, O), np.float32)
        act = np.empty((n, A), np.float32)
        rew = np.empty(n, np.float32)
        done = np.empty(n, np.float32)
        logp = np.empty(n, np.float32)
        for i, s in enumerate(samples):
            obs[i], act[i], rew[i], done[i], logp[i] = np.asarray(s[0]).reshape(-1), s[2], s[3], s[5], s[6]
            obs2[i] = np.asarray(s[4]).reshape(-1)
        self.engine.buffer_add(obs, act, rew, obs2, done, logp)

    @staticmethod
    def _packed_matches(samples, packed):
        """the packed arrays are only trusted while the list still is what the sampler built: a consumer that filtered,
        reordered or replaced tuples (reward shaping, n-step post-processing) gets the tuple walk. Checked on the first,
        middle and last tuple: their action must still BE the packed row (same memory) and their reward the packed value."""
        n = len(samples)
        try:
            for i in {0, n // 2, n - 1}:
                t = samples[i]
                a = t[2]
                if not (isinstance(a, np.ndarray) and a.ctypes.data == packed[1][i].ctypes.data):
                    return False
                if np.float32(t[3]) != packed[2][i] or bool(t[5]) != bool(packed[4][i]):
                    return False
        except Exception:
            return False
        return True

    def sample_batches(self, batch_size: int, n: int):
        """the index rows of the next `n` minibatches: n x `np.random.randint(0, size, batch_size)` -- the calls, and the
        order, of n reference `sample_batch` calls with no add_batch between them (replay_buffer.py:86; the ring size is
        constant, nothing else consumes the NumPy stream). The rows are gathered on the device by
        DSAC_V2_HIP.local_update_group (one graph replay for the n updates)."""
        from dsac_v2_hip import HipBatchGroup

        if batch_size != self.engine.batch:
            raise ValueError("batch_size %d != the engine's minibatch rows %d" % (batch_size, self.engine.batch))
        # ONE call of shape (n, batch): the legacy RandomState fills int64 draws element by element with no buffering between
        # calls, so this is the stream of n calls of `batch` draws (values AND final generator state; tests/test_host_side.py
        # pins it, the trainer-trajectory fixtures compare every index with the reference loop's) at a quarter of the host time
        idxs = np.random.randint(0, self.size, size=(int(n), batch_size))
        return HipBatchGroup(self.engine, idxs)

    def sample_batch(self, batch_size: int):
        from dsac_v2_hip import HipBatch

        idxs = np.random.randint(0, self.size, size=batch_size)  # reference replay_buffer.py:86
        self.engine.gather(idxs)
        return HipBatch(self.engine, idxs)
