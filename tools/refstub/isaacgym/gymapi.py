class _NoOpGym:
    """Every gym.<anything>(...) call is a no-op returning None."""
    def __getattr__(self, name):
        def _f(*a, **k):
            return None
        return _f


def acquire_gym():
    return _NoOpGym()


class CoordinateSpace:
    LOCAL_SPACE = 1
    ENV_SPACE = 0
    GLOBAL_SPACE = 2


class Vec3:
    def __init__(self, x=0., y=0., z=0.):
        self.x, self.y, self.z = x, y, z


class Transform:
    def __init__(self, p=None, r=None):
        self.p, self.r = p, r


class SimParams:
    def __init__(self):
        self.dt = 0.005
        self.use_gpu_pipeline = False


class _Bag:
    def __init__(self, *a, **k):
        pass


PlaneParams = HeightFieldParams = TriangleMeshParams = AssetOptions = CameraProperties = _Bag
SIM_PHYSX = 1
KEY_ESCAPE = 0
KEY_V = 1
