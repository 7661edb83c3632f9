"""ctypes mirror of include/lsim.h, generated at import time by parsing the header.

The header is the single source of truth for the C-ABI (struct layouts, enums,
#defines); parsing it here means a field added to `lsim_config` can never silently
disagree with the Python side.  `lsim_sizeof_config()` / `lsim_sizeof_model()` are
checked against these mirrors when a library is loaded (see `check_abi`).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
HEADER_PATH = os.path.join(REPO_ROOT, "include", "lsim.h")

_CTYPES = {
    "float": ctypes.c_float, "double": ctypes.c_double,
    "int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32,
    "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "int16_t": ctypes.c_int16, "uint8_t": ctypes.c_uint8, "int": ctypes.c_int,
}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _parse(text, overrides=None):
    """`overrides`: values for the header's #ifndef-guarded defines, as a build with -DNAME=value sees them (test-only oracle variants)"""
    text = _strip_comments(text)
    defines = {}
    for m in re.finditer(r"^\s*#define\s+(\w+)\s+(.+?)\s*$", text, flags=re.M):
        name, expr = m.group(1), m.group(2)
        expr = re.sub(r"(\d+)u\b", r"\1", expr)
        if overrides and name in overrides:
            defines[name] = int(overrides[name])
            continue
        try:
            defines[name] = int(eval(expr, {}, dict(defines)))  # only integer arithmetic on earlier defines
        except Exception:
            pass
    enums = {}
    for m in re.finditer(r"enum\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        val = -1
        members = {}
        for item in m.group(2).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = item.split("=")
                val = int(eval(v, {}, {**defines, **members}))
                members[k.strip()] = val
            else:
                val += 1
                members[item] = val
        enums[m.group(1)] = members
        defines.update(members)
    for m in re.finditer(r"^\s*#define\s+(\w+)\s+(.+?)\s*$", text, flags=re.M):   # second pass: defines that use enum members
        name, expr = m.group(1), re.sub(r"(\d+)u\b", r"\1", m.group(2))
        if name not in defines:
            try:
                defines[name] = int(eval(expr, {}, dict(defines)))
            except Exception:
                pass
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            if stmt.startswith("const "):
                stmt = stmt[len("const "):]
            ty, rest = stmt.split(" ", 1)
            if ty.endswith("*"):                         # pointer members (device pointers) bind as void*
                base = ctypes.c_void_p
            else:
                base = structs[ty] if ty in structs else _CTYPES[ty]
            for decl in rest.split(","):
                decl = decl.strip()
                dm = re.match(r"(\w+)((?:\[[^\]]+\])*)$", decl)
                name, dims = dm.group(1), re.findall(r"\[([^\]]+)\]", dm.group(2))
                ct = base
                for d in reversed(dims):
                    ct = ct * int(eval(d, {}, defines))
                fields.append((name, ct))
        structs[m.group(3)] = type(m.group(3), (ctypes.Structure,), {"_fields_": fields})
    return defines, enums, structs


with open(HEADER_PATH) as _f:
    DEFINES, ENUMS, STRUCTS = _parse(_f.read())

LsimConfig = STRUCTS["lsim_config"]
LsimRobotModel = STRUCTS["lsim_robot_model"]
LsimBody = STRUCTS["lsim_body"]
LsimCollisionPoint = STRUCTS["lsim_collision_point"]
LsimRolloutStorage = STRUCTS["lsim_rollout_storage"]
LsimMlpLayer = STRUCTS["lsim_mlp_layer"]
LsimHimPolicy = STRUCTS["lsim_him_policy"]
LsimWgradPending = STRUCTS["lsim_wgrad_pending"]
LsimAmpDisc = STRUCTS["lsim_amp_disc"]

REWARD_IDS = {k[len("LSIM_R_"):].lower(): v for k, v in ENUMS["lsim_reward_id"].items() if k.startswith("LSIM_R_")}
NUM_REWARD_TERMS = ENUMS["lsim_reward_id"]["LSIM_NUM_REWARD_TERMS"]
REWARD_NAMES = [n for n, _ in sorted(REWARD_IDS.items(), key=lambda kv: kv[1])]
BUFFER_IDS = {k[len("LSIM_BUF_"):].lower(): v for k, v in ENUMS["lsim_buffer_id"].items() if k.startswith("LSIM_BUF_")}
NUM_BUFFERS = ENUMS["lsim_buffer_id"]["LSIM_NUM_BUFFERS"]
RNG_TAGS = {k[len("LSIM_RNG_"):].lower(): v for k, v in ENUMS["lsim_rng_tag"].items()}

DT_F32, DT_I64, DT_U8, DT_I32, DT_I16 = (DEFINES[k] for k in ("LSIM_DT_F32", "LSIM_DT_I64", "LSIM_DT_U8", "LSIM_DT_I32", "LSIM_DT_I16"))
E_INVALID, E_NOMEM, E_HIP, E_UNSUPPORTED, E_ABI = (DEFINES[k] for k in ("LSIM_E_INVALID", "LSIM_E_NOMEM", "LSIM_E_HIP", "LSIM_E_UNSUPPORTED", "LSIM_E_ABI"))
ABI_VERSION = DEFINES["LSIM_ABI_VERSION"]
STEP_SKIP_PHYSICS = DEFINES["LSIM_STEP_SKIP_PHYSICS"]
STEP_NO_RESET = DEFINES["LSIM_STEP_NO_RESET"]
STEP_RECORD_SUBSTEPS = DEFINES["LSIM_STEP_RECORD_SUBSTEPS"]
STEP_TWO_KERNELS = DEFINES["LSIM_STEP_TWO_KERNELS"]
STEP_FLAT_PRIORITY = DEFINES["LSIM_STEP_FLAT_PRIORITY"]
STATS = {k[len("LSIM_STATS_"):].lower(): v for k, v in DEFINES.items() if k.startswith("LSIM_STATS_")}


def structs_for(overrides):
    """struct mirrors of a library compiled with -DNAME=value overrides of the header's guarded defines (oracle variants only)"""
    with open(HEADER_PATH) as f:
        return _parse(f.read(), overrides)[2]


def declared_functions():
    """Names of every function the header declares (used by the 'exports every symbol' test)."""
    with open(HEADER_PATH) as f:
        text = _strip_comments(f.read())
    return sorted(set(re.findall(r"\b(lsim_\w+)\s*\(", text)) - {"lsim_sim"})


def check_abi(lib, prefix="lsim", structs=None):
    """Raise if the loaded library was built against a different struct layout."""
    structs = structs or STRUCTS
    for what, struct in (("config", structs["lsim_config"]), ("model", structs["lsim_robot_model"])):
        fn = getattr(lib, f"{prefix}_sizeof_{what}")
        fn.restype = ctypes.c_int
        got = fn()
        if got != ctypes.sizeof(struct):
            raise RuntimeError(f"ABI mismatch: {prefix}_sizeof_{what}() = {got}, python mirror = {ctypes.sizeof(struct)}; rebuild the library")
