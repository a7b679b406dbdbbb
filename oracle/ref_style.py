"""Reference-STYLE scalar Python loop -- TEST INFRASTRUCTURE / CPU baseline only.

A second, independent restatement of the path SURVEY.md section 3.2 describes: one
Python env object per context, stacked the way the reference stacks them
(CARL-style wrapper -> TimeLimit-style wrapper -> env object), scalar ``math``
arithmetic on Python floats, one ``np.array(..., float32)`` allocation and one
``{"obs", "context"}`` dict rebuild per step, ``info["context_id"]``
(carl/envs/carl_env.py:321-342, carl/envs/gymnasium/carl_gymnasium_env.py:75-77).
gymnasium itself is not importable here; its step bodies are restated from the
published 0.29.1 source [upstream-memory] (SURVEY.md section 8a, E-CP .. E-MCC).

Used (a) as ``bench.py``'s ``cpu_baseline`` (kind "port"), (b) to cross-check the
C oracle on small cases.  PARITY UNPINNED, like the C oracle.
"""
from __future__ import annotations

import math

import numpy as np

from . import oracle as O


class _CartPole:
    max_episode_steps = 500

    def __init__(self):
        self.gravity = 9.8
        self.masscart = 1.0
        self.masspole = 0.1
        self.total_mass = self.masspole + self.masscart  # derived ONCE (Quirk C1)
        self.length = 0.5
        self.polemass_length = self.masspole * self.length  # derived ONCE
        self.force_mag = 10.0
        self.tau = 0.02
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        self.state = None
        self.steps_beyond_terminated = None

    def reset(self):
        self.steps_beyond_terminated = None

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if action == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + self.polemass_length * theta_dot**2 * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - self.masspole * costheta**2 / self.total_mass)
        )
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = (x, x_dot, theta, theta_dot)
        terminated = bool(
            x < -self.x_threshold
            or x > self.x_threshold
            or theta < -self.theta_threshold_radians
            or theta > self.theta_threshold_radians
        )
        if not terminated:
            reward = 1.0
        elif self.steps_beyond_terminated is None:
            self.steps_beyond_terminated = 0
            reward = 1.0
        else:
            self.steps_beyond_terminated += 1
            reward = 0.0
        return np.array(self.state, dtype=np.float32), reward, terminated, False, {}


class _Pendulum:
    max_episode_steps = 200

    def __init__(self):
        self.max_speed = 8
        self.max_torque = 2.0
        self.dt = 0.05
        self.g = 10.0
        self.m = 1.0
        self.l = 1.0
        self.state = None

    def reset(self):
        pass

    def step(self, u):
        th, thdot = float(self.state[0]), float(self.state[1])
        g, m, l, dt = self.g, self.m, self.l, self.dt
        u = float(min(max(float(u[0]), -self.max_torque), self.max_torque))
        an = ((th + math.pi) % (2 * math.pi)) - math.pi
        costs = an**2 + 0.1 * thdot**2 + 0.001 * (u**2)
        newthdot = thdot + (3 * g / (2 * l) * math.sin(th) + 3.0 / (m * l**2) * u) * dt
        newthdot = min(max(newthdot, -self.max_speed), self.max_speed)
        newth = th + newthdot * dt
        self.state = np.array([newth, newthdot])
        obs = np.array([math.cos(newth), math.sin(newth), newthdot], dtype=np.float32)
        return obs, -costs, False, False, {}


class _Acrobot:
    max_episode_steps = 500
    dt = 0.2

    def __init__(self):
        self.LINK_LENGTH_1 = 1.0
        self.LINK_LENGTH_2 = 1.0
        self.LINK_MASS_1 = 1.0
        self.LINK_MASS_2 = 1.0
        self.LINK_COM_POS_1 = 0.5
        self.LINK_COM_POS_2 = 0.5
        self.LINK_MOI = 1.0
        self.MAX_VEL_1 = 4 * math.pi
        self.MAX_VEL_2 = 9 * math.pi
        self.torque_noise_max = 0.0
        self.state = None

    def reset(self):
        pass

    def _dsdt(self, y):
        m1, m2 = self.LINK_MASS_1, self.LINK_MASS_2
        l1 = self.LINK_LENGTH_1
        lc1, lc2 = self.LINK_COM_POS_1, self.LINK_COM_POS_2
        I1 = I2 = self.LINK_MOI
        g = 9.8
        theta1, theta2, dtheta1, dtheta2, a = y
        cos, sin, pi = math.cos, math.sin, math.pi
        d1 = m1 * lc1**2 + m2 * (l1**2 + lc2**2 + 2 * l1 * lc2 * cos(theta2)) + I1 + I2
        d2 = m2 * (lc2**2 + l1 * lc2 * cos(theta2)) + I2
        phi2 = m2 * lc2 * g * cos(theta1 + theta2 - pi / 2.0)
        phi1 = (
            -m2 * l1 * lc2 * dtheta2**2 * sin(theta2)
            - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * sin(theta2)
            + (m1 * lc1 + m2 * l1) * g * cos(theta1 - pi / 2)
            + phi2
        )
        ddtheta2 = (
            a + d2 / d1 * phi1 - m2 * l1 * lc2 * dtheta1**2 * sin(theta2) - phi2
        ) / (m2 * lc2**2 + I2 - d2**2 / d1)
        ddtheta1 = -(d2 * ddtheta2 + phi1) / d1
        return np.array([dtheta1, dtheta2, ddtheta1, ddtheta2, 0.0])

    def step(self, a):
        torque = float(a - 1)
        y0 = np.append(np.asarray(self.state, dtype=np.float64), torque)
        dt = self.dt
        dt2 = dt / 2.0
        k1 = self._dsdt(y0)
        k2 = self._dsdt(y0 + dt2 * k1)
        k3 = self._dsdt(y0 + dt2 * k2)
        k4 = self._dsdt(y0 + dt * k3)
        ns = (y0 + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4))[:4]
        for i in (0, 1):
            x = ns[i]
            while x > math.pi:
                x = x - 2 * math.pi
            while x < -math.pi:
                x = x + 2 * math.pi
            ns[i] = x
        ns[2] = min(max(ns[2], -self.MAX_VEL_1), self.MAX_VEL_1)
        ns[3] = min(max(ns[3], -self.MAX_VEL_2), self.MAX_VEL_2)
        self.state = ns
        terminated = bool(-math.cos(ns[0]) - math.cos(ns[1] + ns[0]) > 1.0)
        reward = -1.0 if not terminated else 0.0
        obs = np.array(
            [math.cos(ns[0]), math.sin(ns[0]), math.cos(ns[1]), math.sin(ns[1]), ns[2], ns[3]],
            dtype=np.float32,
        )
        return obs, reward, terminated, False, {}


class _MountainCar:
    max_episode_steps = 200

    def __init__(self):
        self.min_position = -1.2
        self.max_position = 0.6
        self.max_speed = 0.07
        self.goal_position = 0.5
        self.goal_velocity = 0
        self.force = 0.001
        self.gravity = 0.0025
        self.state = None

    def reset(self):
        pass

    def step(self, action):
        position, velocity = self.state
        velocity += (action - 1) * self.force + math.cos(3 * position) * (-self.gravity)
        velocity = min(max(velocity, -self.max_speed), self.max_speed)
        position += velocity
        position = min(max(position, self.min_position), self.max_position)
        if position == self.min_position and velocity < 0:
            velocity = 0
        terminated = bool(position >= self.goal_position and velocity >= self.goal_velocity)
        self.state = (position, velocity)
        return np.array(self.state, dtype=np.float32), -1.0, terminated, False, {}


class _MountainCarContinuous:
    max_episode_steps = 999

    def __init__(self):
        self.min_action = -1.0
        self.max_action = 1.0
        self.min_position = -1.2
        self.max_position = 0.6
        self.max_speed = 0.07
        self.goal_position = 0.45
        self.goal_velocity = 0
        self.power = 0.0015
        self.state = None

    def reset(self):
        pass

    def step(self, action):
        position = float(self.state[0])
        velocity = float(self.state[1])
        force = min(max(float(action[0]), self.min_action), self.max_action)
        velocity += force * self.power - 0.0025 * math.cos(3 * position)
        if velocity > self.max_speed:
            velocity = self.max_speed
        if velocity < -self.max_speed:
            velocity = -self.max_speed
        position += velocity
        if position > self.max_position:
            position = self.max_position
        if position < self.min_position:
            position = self.min_position
        if position == self.min_position and velocity < 0:
            velocity = 0
        terminated = bool(position >= self.goal_position and velocity >= self.goal_velocity)
        reward = 0
        if terminated:
            reward = 100.0
        reward -= math.pow(float(action[0]), 2) * 0.1
        self.state = np.array([position, velocity], dtype=np.float32)
        return self.state, reward, terminated, False, {}


_ENVS = {O.CARTPOLE: _CartPole, O.PENDULUM: _Pendulum, O.ACROBOT: _Acrobot,
         O.MOUNTAINCAR: _MountainCar, O.MOUNTAINCAR_CONT: _MountainCarContinuous}


class _TimeLimit:
    """gymnasium.wrappers.TimeLimit as applied by gymnasium.make."""

    def __init__(self, env, max_episode_steps):
        self.env = env
        self.unwrapped = env
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        observation, reward, terminated, truncated, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            truncated = True
        return observation, reward, terminated, truncated, info

    def reset(self):
        self._elapsed_steps = 0
        return self.env.reset()


class RefStyleEnv:
    """One contextual env the way the reference builds it (scalar, host only).

    ``contexts`` is a dict id -> {feature: value}; selection is round robin
    (carl/context/selection.py:116-122); ``_update_context`` is the setattr loop of
    carl_gymnasium_env.py:75-77; reset applies the family's CARL init distribution
    with uniforms ``u`` supplied by the caller (so a test can feed the engine's
    Philox words) or drawn from a NumPy generator.
    """

    def __init__(self, family: int, contexts: dict | None = None, seed: int = 0):
        self.family = family
        base = _ENVS[family]()
        self.env = _TimeLimit(base, base.max_episode_steps)
        names = O.feature_names(family)
        default = dict(zip(names, O.default_row(family).tolist()))
        if contexts is None:
            contexts = {0: dict(default)}
        self.contexts = {k: {**default, **v} for k, v in contexts.items()}
        self.keys = list(self.contexts.keys())
        self.context_id = None
        self.context = None
        self.obs_context_features = names
        self.rng = np.random.default_rng(seed)

    def _update_context(self):
        for k, v in self.context.items():
            setattr(self.env.unwrapped, k, v)

    def _wrap(self, state):
        return {
            "obs": state,
            "context": {k: v for k, v in self.context.items() if k in self.obs_context_features},
        }

    def reset(self, u=None):
        last = self.context_id
        self.context_id = 0 if self.context_id is None else (self.context_id + 1) % len(self.keys)
        self.context = self.contexts[self.keys[self.context_id]]
        if self.context_id != last:
            self._update_context()
        self.env.reset()
        if u is None:
            u = self.rng.random(4)
        c, f = self.context, self.family
        e = self.env.unwrapped
        if f == O.CARTPOLE:
            lo, hi = c["initial_state_lower"], c["initial_state_upper"]
            e.state = tuple(lo + (hi - lo) * float(u[i]) for i in range(4))
            obs = np.array(e.state, dtype=np.float32)
        elif f == O.PENDULUM:
            th = c["initial_angle_max"] * float(u[0])
            thd = c["initial_velocity_max"] * float(u[1])
            e.state = np.array([th, thd], dtype=np.float32)
            obs = np.array([math.cos(th), math.sin(th), thd], dtype=np.float32)
        elif f == O.ACROBOT:
            alo, ahi = c["INITIAL_ANGLE_LOWER"], c["INITIAL_ANGLE_UPPER"]
            vlo, vhi = c["INITIAL_VELOCITY_LOWER"], c["INITIAL_VELOCITY_UPPER"]
            e.state = np.array([alo + (ahi - alo) * float(u[0]), alo + (ahi - alo) * float(u[1]),
                                vlo + (vhi - vlo) * float(u[2]), vlo + (vhi - vlo) * float(u[3])])
            s = e.state
            obs = np.array([math.cos(s[0]), math.sin(s[0]), math.cos(s[1]), math.sin(s[1]), s[2], s[3]],
                           dtype=np.float32)
        else:
            plo, phi = c["min_position_start"], c["max_position_start"]
            vlo, vhi = c["min_velocity_start"], c["max_velocity_start"]
            e.state = np.array([plo + (phi - plo) * float(u[0]), vlo + (vhi - vlo) * float(u[1])])
            if f == O.MOUNTAINCAR:
                e.state = tuple(e.state.tolist())
            obs = np.array(e.state, dtype=np.float32)
        return self._wrap(obs), {"context_id": self.context_id}

    def step(self, action):
        state, reward, terminated, truncated, info = self.env.step(action)
        state = self._wrap(state)
        info["context_id"] = self.context_id
        return state, reward, terminated, truncated, info


def random_action(family: int, rng: np.random.Generator):
    if family == O.PENDULUM:
        return np.array([rng.uniform(-2, 2)], dtype=np.float32)
    if family == O.MOUNTAINCAR_CONT:
        return np.array([rng.uniform(-1, 1)], dtype=np.float32)
    return int(rng.integers(0, 2 if family == O.CARTPOLE else 3))


def time_loop(family: int, contexts: dict, n_steps_per_env: int, seed: int = 1) -> tuple[int, float]:
    """The reference-style loop over envs: returns (env_steps, seconds). Auto-reset
    on done, like the engine's default."""
    import time

    rng = np.random.default_rng(seed)
    envs = [RefStyleEnv(family, {k: v}, seed=seed + i) for i, (k, v) in enumerate(contexts.items())]
    for e in envs:
        e.reset()
    acts = [random_action(family, rng) for _ in range(256)]
    t0 = time.perf_counter()
    n = 0
    for e in envs:
        for t in range(n_steps_per_env):
            _, _, term, trunc, _ = e.step(acts[(n + t) & 255])
            if term or trunc:
                e.reset()
        n += n_steps_per_env
    return n, time.perf_counter() - t0
