"""The few glm 0.9.9.8 host-side matrix functions the reference's scene front-end uses
(assets/json_parser.cpp:40-95), in float32, column-major: M[col, row] like glm's m[col][row].

Only the host side uses these (scene flattening); the flattened arrays are what the render core and
the parity oracle both consume, so last-ulp differences from glm here cannot break parity."""
import math

import numpy as np

F = np.float32


def identity():
    return np.eye(4, dtype=F)


def translate(v):
    """glm::translate(vec3): Result[3] = m[0]*v0 + m[1]*v1 + m[2]*v2 + m[3] on the identity."""
    m = identity()
    m[3, 0:3] = np.asarray(v, dtype=F)
    return m


def scale(v):
    """glm::scale(vec3) (a scalar means uniform scale, json_parser.cpp:47-52)."""
    v = np.asarray(v, dtype=F)
    if v.ndim == 0:
        v = np.array([v, v, v], dtype=F)
    m = identity()
    m[0, 0], m[1, 1], m[2, 2] = v
    return m


def rotate(angle_rad, axis):
    """glm::rotate(angle, axis) on the identity."""
    a = F(angle_rad)
    c, s = F(math.cos(float(a))), F(math.sin(float(a)))
    axis = np.asarray(axis, dtype=F)
    axis = axis / F(math.sqrt(float(np.dot(axis, axis))))
    temp = (F(1) - c) * axis
    r = identity()
    r[0, 0] = c + temp[0] * axis[0]
    r[0, 1] = temp[0] * axis[1] + s * axis[2]
    r[0, 2] = temp[0] * axis[2] - s * axis[1]
    r[1, 0] = temp[1] * axis[0] - s * axis[2]
    r[1, 1] = c + temp[1] * axis[1]
    r[1, 2] = temp[1] * axis[2] + s * axis[0]
    r[2, 0] = temp[2] * axis[0] + s * axis[1]
    r[2, 1] = temp[2] * axis[1] - s * axis[0]
    r[2, 2] = c + temp[2] * axis[2]
    return r


def _normalize(v):
    v = np.asarray(v, dtype=F)
    return v * (F(1) / F(math.sqrt(float(F(v[0] * v[0] + v[1] * v[1]) + F(v[2] * v[2])))))


def look_at(frm, at, up):
    """The 'from/at/up' transform command, json_parser.cpp:57-70 (columns left, up, dir, from)."""
    frm, at, up = (np.asarray(x, dtype=F) for x in (frm, at, up))
    d = _normalize(frm - at)
    left = _normalize(np.cross(up, d).astype(F))
    new_up = _normalize(np.cross(d, left).astype(F))
    m = identity()
    m[0, 0:3], m[1, 0:3], m[2, 0:3], m[3, 0:3] = left, new_up, d, frm
    return m


def matmul(a, b):
    """glm mat4 * mat4: column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3."""
    r = np.zeros((4, 4), dtype=F)
    for j in range(4):
        r[j] = ((a[0] * b[j, 0] + a[1] * b[j, 1]) + a[2] * b[j, 2]) + a[3] * b[j, 3]
    return r


def compose(commands):
    """A transform array is applied left to right as elem * mat (json_parser.cpp:85-88)."""
    m = identity()
    for c in commands:
        m = matmul(c, m)
    return m


def quat_from_matrix(m):
    """Rotation part of glm::decompose (gtx/matrix_decompose.inl) for a matrix without scale/skew.
    Returns (w, x, y, z)."""
    row = [np.asarray(m[i, 0:3], dtype=np.float64) for i in range(3)]
    trace = row[0][0] + row[1][1] + row[2][2]
    q = [0.0, 0.0, 0.0, 0.0]  # x y z w
    if trace > 0:
        root = math.sqrt(trace + 1.0)
        q[3] = 0.5 * root
        root = 0.5 / root
        q[0] = root * (row[1][2] - row[2][1])
        q[1] = root * (row[2][0] - row[0][2])
        q[2] = root * (row[0][1] - row[1][0])
    else:
        nxt = [1, 2, 0]
        i = 0
        if row[1][1] > row[0][0]:
            i = 1
        if row[2][2] > row[i][i]:
            i = 2
        j, k = nxt[i], nxt[nxt[i]]
        root = math.sqrt(row[i][i] - row[j][j] - row[k][k] + 1.0)
        q[i] = 0.5 * root
        root = 0.5 / root
        q[j] = root * (row[i][j] + row[j][i])
        q[k] = root * (row[i][k] + row[k][i])
        q[3] = root * (row[j][k] - row[k][j])
    return (F(q[3]), F(q[0]), F(q[1]), F(q[2]))
