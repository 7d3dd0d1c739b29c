"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32) of the reference's DSAC_V1 update with the CNN approximators
(value_func_type = policy_func_type = "CNN": example_train/dsacv1_cnn_carracing_offasync.py; dsac_v1.py:140-279 over
networks/cnn.py:151-240,383-461).

The update is oracle/dsac_v1_oracle.py's (one critic, fixed TD_bound, both critic losses); the approximators are
oracle/dsact_oracle_cnn.py's (conv stack -> flatten -> separate `mean` / `log_std` MLPs, `cnn_shared` False). Pinned against the
live reference by tests/test_oracle_vs_reference.py::test_v1_cnn_bit_exact_vs_live_reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from typing import Dict

from .dsac_v1_oracle import DsacV1Oracle
from .dsact_oracle_cnn import CONV_TYPES, DsactCnnOracle, conv_out_hw


class DsacV1CnnOracle(DsacV1Oracle):
    def __init__(self, cfg: Dict, state_dict=None):
        self.ks, self.ch, self.st, hid = CONV_TYPES[cfg["conv_type"]]
        assert list(cfg["hidden"]) == list(hid)
        C, H, W = cfg["obs_dim"]
        self.n_conv = len(self.ks)
        oh, ow = conv_out_hw(H, W, self.ks, self.st)[-1]
        self.feat_dim = self.ch[-1] * oh * ow
        self.keep_conv, self.conv_acts = False, []
        self.relu_masks, self.mask_input, self.kinks = {}, None, []
        super().__init__(cfg, state_dict)

    # the approximators and their checkpoint names: exactly DsactCnnOracle's
    _new_net = DsactCnnOracle._new_net
    _new_q_params = DsactCnnOracle._new_q_params
    _new_pi_params = DsactCnnOracle._new_pi_params
    _split = DsactCnnOracle._split
    _conv = DsactCnnOracle._conv
    _names = DsactCnnOracle._names

    def _pi(self, obs, params):
        return DsactCnnOracle._pi(self, obs, params)

    def _q(self, obs, act, params):
        return DsactCnnOracle._q(self, obs, act, params)
