"""lphash_amd -- MI355X-native LPHash query engine (query-p / query-u hot path).

C ABI: include/lphq.h (liblphq.so).  This package is the Python host mirror of the reference's MPHF
interface over that ABI.
"""
from ._lib import BBHASH, INVALID, PARTITIONED, PTHASH, UNPARTITIONED, LphqError, lib  # noqa: F401
from .api import BbHash, Mphf, MphfAlt, PtHash, build, build_bbhash, build_pthash, build_stream  # noqa: F401
