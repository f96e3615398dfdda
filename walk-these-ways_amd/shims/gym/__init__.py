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


class spaces:  # noqa: N801
    pass
