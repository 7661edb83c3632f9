def parse_device_str(s):
    if ':' in s:
        t, i = s.split(':')
        return t, int(i)
    return s, 0


def parse_arguments(*a, **k):
    raise RuntimeError("stub")


class WireframeSphereGeometry:
    def __init__(self, *a, **k):
        pass


def draw_lines(*a, **k):
    pass
