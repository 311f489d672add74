"""
Vectorised host-side environments used by the tests, the bench and the solve() mirror.

TestMDP        restatement of the reference's own image-observation test MDP (test/test_env.jl:10-87):
               state = (last `stack`=4 actions in {1,2,3}, t); observation = stack of `o_stack` fixed random integer
               images / 255 (:54-60); reward [-0.1, 0, +0.1][sp[end]] times -10 if s[end] == 2 (:77-83); terminal when
               t >= max_time (:85-87); 4 actions, action 4 repeats the previous element (:66-75); discount 0.99.
               Known answer (test/test_env.jl:7-8): optimal return 2.1, optimal policy [2,1,2,1,3].
               TestMDP((84,84),4,6) yields exactly the 84x84x4 observation of BASELINE configs 2/3/5.
SimpleGridWorld  POMDPModels.SimpleGridWorld defaults (third-party; recalled): 10x10 grid, 4 actions,
               rewards (4,3)=-10 (4,6)=-5 (9,3)=+10 (8,8)=+3 which are terminal, 70 % intended-move probability,
               discount 0.95, observation Float32[x, y].

Actions are 0-based here.  `n` environments step in lock-step as NumPy arrays (BASELINE config 3 shards 256 of
them over 8 ranks).
"""
from __future__ import annotations

import numpy as np


class TestMDP:
    __test__ = False  # not a pytest class

    def __init__(self, shape=(6,), stack=4, max_time=6, discount=0.99, n=1, seed=7, u8=False):
        rng = np.random.default_rng(seed)
        self.shape = tuple(shape)
        self.stack = 4              # hard-coded field (constructor quirk, test/test_env.jl:31)
        self.o_stack = stack        # the `stack` ARGUMENT becomes o_stack
        self.max_time = max_time
        self.discount = discount
        img_shape = self.shape[::-1]  # Julia (W,H) -> C order (H,W)
        self.images = np.stack([rng.integers(1, 51, img_shape), rng.integers(100, 151, img_shape),
                                rng.integers(150, 201, img_shape)]).astype(np.uint8)   # bad, normal, good (:26-28)
        self.rewards = np.array([-0.1, 0.0, 0.1], np.float32)
        self.n = n
        self.u8 = u8
        self.rng = np.random.default_rng(seed + 1)
        self.n_actions = 4
        self.obs_shape = (self.o_stack,) + img_shape
        self.reset()

    def reset(self, mask=None):
        if mask is None:
            self.s = np.ones((self.n, self.stack), np.int32)
            self.t = np.ones(self.n, np.int32)
        else:
            self.s[mask] = 1
            self.t[mask] = 1

    def observe(self):
        # obs[.., i] = observations[s[end-i+1]]  (test/test_env.jl:56-58)
        sel = self.s[:, ::-1][:, :self.o_stack] - 1            # (n, o_stack)
        o = self.images[sel]                                    # (n, o_stack, H, W) uint8
        return o if self.u8 else o.astype(np.float32) / np.float32(255.0)

    def terminated(self):
        return self.t >= self.max_time

    def act(self, a):
        a = np.asarray(a)
        was_second = self.s[:, -1] == 2
        s_new = np.roll(self.s, -1, axis=1)
        s_new[:, -1] = np.where(a < 3, a + 1, s_new[:, -2])
        r = self.rewards[s_new[:, -1] - 1] * np.where(was_second, np.float32(-10), np.float32(1))
        self.s, self.t = s_new, self.t + 1
        return r.astype(np.float32)


class SimpleGridWorld:
    def __init__(self, size=(10, 10), n=1, seed=0, tprob=0.7, discount=0.95):
        self.size = size
        self.reward_cells = {(4, 3): -10.0, (4, 6): -5.0, (9, 3): 10.0, (8, 8): 3.0}
        self.tprob, self.discount = tprob, discount
        self.n, self.n_actions, self.obs_shape = n, 4, (2,)
        self.dirs = np.array([[0, 1], [0, -1], [-1, 0], [1, 0]], np.int32)   # up, down, left, right
        self.rng = np.random.default_rng(seed)
        self.rmap = np.zeros((size[0] + 1, size[1] + 1), np.float32)
        for (x, y), v in self.reward_cells.items():
            self.rmap[x, y] = v
        self.reset()

    def reset(self, mask=None):
        new = np.stack([self.rng.integers(1, self.size[0] + 1, self.n), self.rng.integers(1, self.size[1] + 1, self.n)], 1).astype(np.int32)
        if mask is None:
            self.pos = new
            self.done = np.zeros(self.n, bool)
        else:
            self.pos[mask] = new[mask]
            self.done[mask] = False

    def observe(self):
        return self.pos.astype(np.float32)

    def terminated(self):
        return self.done

    def act(self, a):
        a = np.asarray(a)
        r = self.rmap[self.pos[:, 0], self.pos[:, 1]].copy()     # reward for acting from a reward cell, then terminal
        at_reward = r != 0
        rnd = self.rng.random(self.n) < self.tprob
        other = self.rng.integers(0, 3, self.n)
        eff = np.where(rnd, a, (a + 1 + other) % 4)
        new = self.pos + self.dirs[eff]
        inb = (new[:, 0] >= 1) & (new[:, 0] <= self.size[0]) & (new[:, 1] >= 1) & (new[:, 1] <= self.size[1])
        self.pos = np.where((inb & ~at_reward)[:, None], new, self.pos)
        self.done = at_reward
        return r.astype(np.float32)
