"""Only the names user scripts touch."""
SIM_PHYSX = "SIM_PHYSX"
IMAGE_COLOR = 0


class _Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class SimParams(_Bag):
    def __init__(self):
        super().__init__(dt=1 / 60., substeps=2, up_axis=1, use_gpu_pipeline=True, gravity=[0., 0., -9.81], physx=_Bag())


def Vec3(x=0., y=0., z=0.):
    return (x, y, z)
