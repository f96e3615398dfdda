"""Stand-in for OpenAI gym's two base classes used by the env surface (absent from this image)."""


class Env:
    def close(self):
        pass


class Wrapper(Env):
    """Attribute access falls through to the wrapped env; assignment stays on the wrapper (gym semantics,
    SURVEY.md App. D4)."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return getattr(self.env, "unwrapped", self.env)

    # gym.Wrapper's explicit delegations: subclasses reach them through super() (the reference HistoryWrapper calls
    # super().reset(), history_wrapper.py:38 — attribute fall-through does not serve super objects)
    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def render(self, mode="human", **kwargs):
        return self.env.render(mode, **kwargs)

    def close(self):
        return self.env.close()

    def seed(self, seed=None):
        return self.env.seed(seed)


class spaces:  # noqa: N801
    pass
