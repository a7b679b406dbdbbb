"""Branch-aware per-step parity of the HIP Brax kernel against the float64 restatement (oracle/brax_spring.c).

north_star's bar is "within 1e-5 fp32 (bit-exact for discrete done flags and context indexing)".  An env step of
the spring pipeline is a smooth function of the state EXCEPT where a discrete decision flips: a collision sphere
delivers an impulse only while it penetrates AND approaches (vn < 0), and termination is a threshold on the root
height.  Two correct implementations of the same arithmetic, one in float32 and one in float64, therefore agree to
rounding on every lane that took the same decisions, and differ by O(impulse) on the few lanes where a contact
switched inside the rounding interval -- exactly like the threshold-edge done flags of the classic-control tests
(tests/test_gpu_parity.py), which are excluded and counted.  Both sides record their decisions (kernel:
carl_step_io_t::branch_sig; oracle: obx_engine_step's branch_sig): word 0 hashes which spheres delivered an
impulse in which substep, word 1 which joint range limits were active (the limit spring starts at zero force, so
a flip there is continuous: reported, not excluded).

`run` steps both sides from the SAME state (the oracle restarts every env step from the engine's float64 state
view), and returns the per-lane errors of the lanes whose contact hash and `terminated` flag agree, plus the
shares of the excluded lanes.  Tests assert max <= 1e-5 on the former and bound the latter.
"""
import numpy as np
import torch


def rel_err(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want) / (1.0 + np.abs(want))


class Parity:
    def __init__(self):
        self.err = []          # per lane-step error (obs entries and reward), agreeing lanes only
        self.err_all = []      # every lane-step (diagnostics)
        self.lane_steps = 0
        self.contact_mismatch = 0
        self.limit_mismatch = 0
        self.flag_mismatch = 0
        self.worst = None      # (error, step, lane, column) of the worst agreeing entry

    def shares(self):
        n = max(self.lane_steps, 1)
        return {"contact": self.contact_mismatch / n, "limit": self.limit_mismatch / n, "terminated": self.flag_mismatch / n}

    def summary(self, label=""):
        e = np.concatenate(self.err) if self.err else np.zeros(1)
        a = np.concatenate(self.err_all) if self.err_all else np.zeros(1)
        sh = self.shares()
        return (f"{label:26s} lane-steps {self.lane_steps:7d}  agreeing: p50 {np.percentile(e, 50):.2e} p99 {np.percentile(e, 99):.2e} "
                f"max {e.max():.2e} | excluded: contact {sh['contact']:.5f} terminated {sh['terminated']:.5f} "
                f"(limit flips, not excluded: {sh['limit']:.5f}) | all lanes: p99 {np.percentile(a, 99):.2e} max {a.max():.2e} "
                f"share>1e-5 {np.mean(a > 1e-5):.5f}")


def step_both(eng, ora, action, par: Parity, t=0, sync_goal=False):
    """One env step on both sides from the engine's state.  Returns (obs, rew, term, trunc, out)."""
    ora.state[:] = eng.state_np()
    if sync_goal:
        ora.goal_pos[:] = eng.goal_pos.t().cpu().numpy()
    obs, rew, term, trunc = eng.step(torch.as_tensor(action))
    out = ora.step(action)
    term_g, trunc_g = term.cpu().numpy() != 0, trunc.cpu().numpy() != 0
    np.testing.assert_array_equal(trunc_g, out.truncated != 0)  # TimeLimit is integer arithmetic: exact
    sig = eng.branch_sig.cpu().numpy().view(np.uint32)
    flag = term_g != (out.terminated != 0)
    contact = sig[:, 0] != ora.branch_sig[:, 0]
    limit = sig[:, 1] != ora.branch_sig[:, 1]
    agree = ~flag & ~contact
    done = (term_g | trunc_g)
    has_final = bool(eng.b.flags & 1)  # auto-reset: the transition's own observation is the terminal one
    if has_final:
        got_obs = np.where(done[:, None], eng.final_obs.cpu().numpy(), obs.cpu().numpy())
        want_obs = np.where(done[:, None], out.final_obs, out.obs)
    else:
        got_obs, want_obs = obs.cpu().numpy(), out.obs
    eo = rel_err(got_obs, want_obs)
    e = np.maximum(eo.max(1), rel_err(rew.cpu().numpy(), out.reward))
    par.err.append(e[agree])
    par.err_all.append(e)
    par.lane_steps += e.size
    par.contact_mismatch += int((contact & ~flag).sum())
    par.limit_mismatch += int(limit.sum())
    par.flag_mismatch += int(flag.sum())
    if agree.any():
        k = int(np.argmax(np.where(agree, e, -1.0)))
        if par.worst is None or e[k] > par.worst[0]:
            par.worst = (float(e[k]), t, k, int(np.argmax(eo[k])))
    if flag.any():  # one side reset its env and the other did not: keep the episode bookkeeping together
        ora.elapsed[:] = eng.elapsed.cpu().numpy()
        ora.episode[:] = eng.episode.cpu().numpy().view(np.uint32)
        ora.ep_return[:] = eng.ep_return.cpu().numpy()
        ora.episodes_done[:] = eng.episodes_done.cpu().numpy()
        ora.n_calls[:] = eng.n_calls.cpu().numpy()
        ora.ctx_idx[:] = eng.ctx_idx.cpu().numpy()
    return obs, rew, term, trunc, out


def assert_parity(par: Parity, label, tol=1e-5, max_excluded=1e-3):
    """north_star's tolerance as a MAXIMUM over every lane-step whose discrete decisions agree; the excluded share
    is bounded and printed."""
    print(par.summary(label), "worst agreeing entry (err, step, lane, column):", par.worst, flush=True)
    e = np.concatenate(par.err)
    sh = par.shares()
    assert sh["contact"] + sh["terminated"] <= max_excluded, sh
    assert e.size >= (1.0 - max_excluded) * par.lane_steps
    assert e.max() <= tol, par.summary(label)
