"""Geometries of tests/golden/dsconv.npz: (in_channels, out_channels, kernel_size, stride, padding, dilation, bias, T)."""
CASES = {
    "default_stride": (8, 12, 3, None, 0, 1, True, 100),
    "same_s1": (8, 12, 3, 1, 1, 1, True, 100),
    "same_s1_d4": (8, 12, 3, 1, 4, 4, True, 100),
    "s2_p1": (8, 12, 3, 2, 1, 1, True, 101),
    "s2_d2_p3_nobias": (6, 5, 5, 2, 3, 2, False, 97),
    "valid_s1": (8, 12, 3, 1, 0, 1, True, 50),
    "overpadded": (4, 4, 3, 1, 5, 1, True, 40),
    "s3_k7_p2": (16, 8, 7, 3, 2, 1, True, 200),
    "long_s4_k16": (32, 16, 16, 4, 6, 1, True, 4000),
}
