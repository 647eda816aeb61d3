"""Constants of the rasterizer path (reference config.py:17-23): 16x16 tiles, VEC6 = 6 packed floats
in upper-triangle order (xx,xy,xz,yy,yz,zz).  The device is whatever torch device the inputs live on;
there is no DEVICE global and no CPU fallback."""
TILE_M = 16
TILE_N = 16
TILE_THREADS = 256
VEC6_LEN = 6
SH_STRIDE = 16
MAX_RENDERED = 1 << 30
