/*
 * utils.h — the rectangle / pose struct API of the collision engine.
 *
 * These PODs are the boundary types of the hot path.  Names and field order
 * follow the reference's struct block (reference utils.cu:74-106; the
 * reference's own utils.h is dead code) so that the reference's
 * .npy rows and host drivers map onto them without repacking:
 *
 *   Position                  {x,y}                       utils.cu:74-77
 *   PositionWithVarAndPoseIdx {x,y,var_idx,pose_idx}      utils.cu:79-84   (= one row of data_in/<k>.npy)
 *   Variance / StdDev         {x,y,theta,width,height}    utils.cu:86-89,106
 *   Pose                      {width,height,theta}        utils.cu:91-94
 *   PoseCPVarAndPoseIdx       {x,y,cp,var_idx,pose_idx}   utils.cu:96-99   (= one row of data_out/<k>.npy)
 *   PoseCPVarAndPoseIdxIdx    {p, idx}                    utils.cu:100-104 (a row + its original position; the
 *                             reference needs it to undo its compaction sort, compute_collision_probability.cu:337-344;
 *                             c2d keeps scene i in slot i, so nothing here produces it — it is published for callers
 *                             that carry the reference's host code over)
 *
 * All fields are IEEE-754 binary32; indices are stored *as float* exactly as
 * the reference does (utils.cu:82-83).  Plain C, usable from C, C++, HIP and
 * (through ctypes) Python.
 */
#ifndef C2D_UTILS_H_
#define C2D_UTILS_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Position {
    float x, y;
} Position;

typedef struct PositionWithVarAndPoseIdx {
    float x, y;
    float var_idx;
    float pose_idx;
} PositionWithVarAndPoseIdx;

typedef struct Variance {
    float x, y, theta, width, height;
} Variance;

typedef Variance StdDev;

typedef struct Pose {
    float width, height, theta;
} Pose;

typedef struct PoseCPVarAndPoseIdx {
    float x, y, cp, var_idx, pose_idx;
} PoseCPVarAndPoseIdx;

typedef struct PoseCPVarAndPoseIdxIdx {
    PoseCPVarAndPoseIdx p;
    int idx;
} PoseCPVarAndPoseIdxIdx;

/* A rectangle is a flat float[8]: x0,y0,x1,y1,x2,y2,x3,y3, counter-clockwise,
 * starting at (-w/2,-h/2) (reference utils.cu:119-130). */
#define C2D_RECT_FLOATS 8

/* Arbitrary convex polygons carry at most this many vertices (BASELINE config 5). */
#define C2D_POLY_KMAX 16

/* create_rect (reference utils.cu:119-130, a __device__ __host__ function the reference's mains
 * call on the host, e.g. compute_collision_probability.cu:240): the 4 counter-clockwise vertices
 * of a w x h box centred at the origin, starting at (-w/2, -h/2). */
static inline void create_rect(float* r, float w, float h)
{
    r[0] = -w / 2;
    r[1] = -h / 2;
    r[2] = w / 2;
    r[3] = -h / 2;
    r[4] = w / 2;
    r[5] = h / 2;
    r[6] = -w / 2;
    r[7] = h / 2;
}

#ifdef __cplusplus
}
#endif

#endif /* C2D_UTILS_H_ */
