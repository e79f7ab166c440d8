"""VideoConfigurations.jl mirrored on the host: the table of TOTAL raster sizes
(blanking included) and the nearest-rate-then-nearest-height lookup the configuration search ends
with (VideoConfigurations.jl:99-124; callers GUI.jl:263,571-574, production/investigate_data.jl:92).

Host-side by design: it is a 80-entry lookup, not a data-parallel path.
"""
import warnings
from dataclasses import dataclass


@dataclass(eq=False)  # the reference's VideoMode is a mutable struct without ==: identity semantics
class VideoMode:
    width: int      # total pixels per line (x_t)
    height: int     # total lines (y_t)
    refresh: float  # Hz


_TABLE = [
    ('PAL TV', 576, 625, 25.0),
    ('640x400 @ 85Hz', 832, 445, 85.0),
    ('720x400 @ 85Hz', 936, 446, 85.0),
    ('640x480 @ 60Hz', 800, 525, 60.0),
    ('640x480 @ 100Hz', 848, 509, 100.0),
    ('640x480 @ 72Hz', 832, 520, 72.0),
    ('640x480 @ 75Hz', 840, 500, 75.0),
    ('640x480 @ 85Hz', 832, 509, 85.0),
    ('768x576 @ 60 Hz', 976, 597, 60.0),
    ('768x576 @ 72 Hz', 992, 601, 72.0),
    ('768x576 @ 75 Hz', 1008, 602, 75.0),
    ('768x576 @ 85 Hz', 1008, 605, 85.0),
    ('768x576 @ 100 Hz', 1024, 611, 100.0),
    ('800x600 @ 56Hz', 1024, 625, 56.0),
    ('800x600 @ 60Hz', 1056, 628, 60.0),
    ('800x600 @ 72Hz', 1040, 666, 72.0),
    ('800x600 @ 75Hz', 1056, 625, 75.0),
    ('800x600 @ 85Hz', 1048, 631, 85.0),
    ('800x600 @ 100Hz', 1072, 636, 100.0),
    ('1024x600 @ 60 Hz', 1312, 622, 60.0),
    ('1024x768i @ 43Hz', 1264, 817, 43.0),
    ('1024x768 @ 60Hz', 1344, 806, 60.0),
    ('1024x768 @ 70Hz', 1328, 806, 70.0),
    ('1024x768 @ 75Hz', 1312, 800, 75.0),
    ('1024x768 @ 85Hz', 1376, 808, 85.0),
    ('1024x768 @ 100Hz', 1392, 814, 100.0),
    ('1024x768 @ 120Hz', 1408, 823, 120.0),
    ('1152x864 @ 60Hz', 1520, 895, 60.0),
    ('1152x864 @ 75Hz', 1600, 900, 75.0),
    ('1152x864 @ 85Hz', 1552, 907, 85.0),
    ('1152x864 @ 100Hz', 1568, 915, 100.0),
    ('1280x768 @ 60 Hz', 1680, 795, 60.0),
    ('1280x800 @ 60 Hz', 1680, 828, 60.0),
    ('1280x960 @ 60Hz', 1800, 1000, 60.0),
    ('1280x960 @ 75Hz', 1728, 1002, 75.0),
    ('1280x960 @ 85Hz', 1728, 1011, 85.0),
    ('1280x960 @ 100Hz', 1760, 1017, 100.0),
    ('1280x1024 @ 60Hz', 1688, 1066, 60.0),
    ('1280x1024 @ 75Hz', 1688, 1066, 75.0),
    ('1280x1024 @ 85Hz', 1728, 1072, 85.0),
    ('1280x1024 @ 100Hz', 1760, 1085, 100.0),
    ('1280x1024 @ 120Hz', 1776, 1097, 120.0),
    ('1368x768 @ 60 Hz', 1800, 795, 60.0),
    ('1400x1050 @ 60Hz', 1880, 1082, 60.0),
    ('1400x1050 @ 72 Hz', 1896, 1094, 72.0),
    ('1400x1050 @ 75 Hz', 1896, 1096, 75.0),
    ('1400x1050 @ 85 Hz', 1912, 1103, 85.0),
    ('1400x1050 @ 100 Hz', 1928, 1112, 100.0),
    ('1440x900 @ 60 Hz', 1904, 932, 60.0),
    ('1440x1050 @ 60 Hz', 1936, 1087, 60.0),
    ('1600x1000 @ 60Hz', 2144, 1035, 60.0),
    ('1600x1000 @ 75Hz', 2160, 1044, 75.0),
    ('1600x1000 @ 85Hz', 2176, 1050, 85.0),
    ('1600x1000 @ 100Hz', 2192, 1059, 100.0),
    ('1600x1024 @ 60Hz', 2144, 1060, 60.0),
    ('1600x1024 @ 75Hz', 2176, 1069, 75.0),
    ('1600x1024 @ 76Hz', 2096, 1070, 76.0),
    ('1600x1024 @ 85Hz', 2176, 1075, 85.0),
    ('1600x1200 @ 60Hz', 2160, 1250, 60.0),
    ('1600x1200 @ 65Hz', 2160, 1250, 65.0),
    ('1600x1200 @ 70Hz', 2160, 1250, 70.0),
    ('1600x1200 @ 75Hz', 2160, 1250, 75.0),
    ('1600x1200 @ 85Hz', 2160, 1250, 85.0),
    ('1600x1200 @ 100 Hz', 2208, 1271, 100.0),
    ('1680x1050 @ 60Hz (reduced blanking)', 1840, 1080, 60.0),
    ('1680x1050 @ 60Hz (non-interlaced)', 2240, 1089, 60.0),
    ('1680x1050 @ 60 Hz', 2256, 1087, 60.0),
    ('1792x1344 @ 60Hz', 2448, 1394, 60.0),
    ('1792x1344 @ 75Hz', 2456, 1417, 75.0),
    ('1856x1392 @ 60Hz', 2528, 1439, 60.0),
    ('1856x1392 @ 75Hz', 2560, 1500, 75.0),
    ('1920x1080 @ 60Hz', 2576, 1125, 60.0),
    ('1920x1080 @ 75Hz', 2608, 1126, 75.0),
    ('1920x1200 @ 60Hz', 2592, 1242, 60.0),
    ('1920x1200 @ 75Hz', 2624, 1253, 75.0),
    ('1920x1440 @ 60Hz', 2600, 1500, 60.0),
    ('1920x1440 @ 75Hz', 2640, 1500, 75.0),
    ('1920x2400 @ 25Hz', 2048, 2434, 25.0),
    ('1920x2400 @ 30Hz', 2044, 2434, 30.0),
    ('2048x1536 @ 60Hz', 2800, 1589, 60.0),
]

# Dict{String,VideoMode} -- VideoConfigurations.jl:12-93 (values are the table's data)
allVideoConfigurations = {name: VideoMode(w, h, r) for name, w, h, r in _TABLE}


def get_refresh_rates(subdict):
    """unique refresh rates in iteration order -- VideoConfigurations.jl:128-130"""
    seen = []
    for m in subdict.values():
        if m.refresh not in seen:
            seen.append(m.refresh)
    return seen


# (height, refresh) pairs of the table that TWO modes of different width share.  For a line count nearest to one of these
# heights the reference returns both entries and @warns (:104-106); its callers then take the first entry of the
# sub-Dict (dict2video, investigate_data.jl:92-97), i.e. whichever comes first in Julia's hash order of the String keys
# -- a property of Base.hash and the Dict's growth history, not of TempestSDR.jl.  Here the order is the table's own
# (VideoConfigurations.jl:12-93, top to bottom).  Everywhere else the result does not depend on any order.
# tests/golden/make_golden.jl dumps Julia's order so that a Julia run can pin the three picks.
AMBIGUOUS_HEIGHT_REFRESH = ((795, 60.0), (1087, 60.0), (1500, 75.0))


def _find_closest_configuration(y_t, d):
    """:99-108 -- every entry whose height is nearest to y_t (squared distance, exact ties kept)"""
    dist = [abs(float(y_t) - m.height) ** 2 for m in d.values()]
    vv = min(dist)
    sub = {k: m for k, m in d.items() if abs(float(y_t) - m.height) ** 2 == vv}
    if len(sub) > 1:
        warnings.warn(f"Several configurations are valid for y_t={y_t} and refresh rate {get_refresh_rates(d)}")
    return sub


def find_closest_configuration(y_t, r):
    """nearest refresh rate first (first minimum wins), then nearest height -- :117-124"""
    rates = get_refresh_rates(allVideoConfigurations)
    d2 = [abs(r - x) ** 2 for x in rates]
    chosen = rates[d2.index(min(d2))]
    sub = {k: m for k, m in allVideoConfigurations.items() if m.refresh == chosen}
    return _find_closest_configuration(y_t, sub)


def find_configuration(video):
    """name of the table entry that IS `video` (identity, as the reference's == on a mutable struct) -- :136-142"""
    for k, m in allVideoConfigurations.items():
        if m is video:
            return k
    return None


def dict2video(subdict):
    """:144-146"""
    return list(subdict.values())[0]


# lag <-> line-count helpers of the GUI (GUI.jl:238-252)
def delay2yt(tau, fv):
    return round(1.0 / (fv * tau))


def yt2index(yt, Fs, fv):
    return round(Fs / (fv * yt))


def yt2delay(yt, fv):
    return 1.0 / (fv * yt)
