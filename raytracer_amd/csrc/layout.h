// HBM data layout of the flattened scene (shared by the host builder and the HIP kernels).
//
// The reference keeps one pointer-based cy::BVHTriMesh per TriObj (objects.cpp:136, triobj_bvh.cpp:13-34) and
// walks the scene graph per ray (IntersectNodeAdapter.h:23-97). Here all (Node,TriObj) pairs are merged into ONE
// 4-wide BVH over world-space boxes with quantised child boxes (triangles of identity nodes are tested in world
// space - every object of cerealBar.xml -, those of transformed nodes in their object space):
//
//   Node4   64 B   origin (3 x f32) + 3 power-of-two scales + 4 x (6 x u8 box) + 4 child references
//   TriRec  48 B   v0, e1 = v1-v0, e2 = v2-v0 (the operands of objects.cpp:146-151) + global triangle id
//   TriAttr 128 B  vertex normals, texture coordinates, object / material id (objects.cpp:181-236) + the record's edges and words
//
// TriRec and TriAttr are stored in BVH (depth-first leaf) order so that neighbouring hits touch neighbouring
// memory; a leaf references up to 4 consecutive records (leaf size 4 as in bvh.SetMesh(this, 4)).
#pragma once
#include <stdint.h>

#define RTX_LEAF_FLAG 0x80000000u
#define RTX_EMPTY_CHILD 0xFFFFFFFFu
#ifndef RTX_MAX_LEAF
#define RTX_MAX_LEAF 2
#endif

// traversal stack budget per lane: STACK_LDS entries in LDS, STACK_GLOBAL more in a global spill slab
#ifndef STACK_LDS
#define STACK_LDS 12
#endif
#define STACK_GLOBAL (128 - STACK_LDS)
#define TRAV_BLOCK 256

// child reference: inner node -> node index; leaf -> LEAF_FLAG | (firstTri << 3) | (count - 1); either one | RTX_REF_TRANS (bit 30,
// traverse.h) when the subtree holds a surface that lets light through
struct Node4 {
    float ox, oy, oz;       // quantisation origin = node box min
    uint8_t ex, ey, ez, nchild;  // biased fp32 exponents of the per-axis scale (scale = 2^(e-127))
    uint32_t child[4];
    uint8_t qlo[3][4];      // [axis][child], box lo = o + q * scale (rounded down)
    uint8_t qhi[3][4];      // box hi (rounded up)
    uint32_t pad[2];        // [0] bit i: child i's subtree holds a surface that lets light through (no TRI_OPAQUE); [1]: any child does
};
static_assert(sizeof(Node4) == 64, "Node4 must be 64 bytes");

// The 8-wide form (RTX_BVH8=1): 128-byte nodes with implicit child addressing, after Ylitie, Karras, Laine, "Efficient
// incoherent ray traversal on GPUs through compressed wide BVHs" (HPG 2017). A node's inner children are consecutive nodes
// from childBase in slot order, the triangles of its leaf children consecutive records from triBase in slot order, so a
// node carries two bases and a few masks instead of eight references, and the traversal stack holds one entry per level:
// the node GROUP {childBase, hit slots still to visit}. Children sit in the slot whose corner of the node they lie towards
// (bit a of the slot set = high side on axis a), so "slot XOR ray octant" orders them front to back without a sort.
#ifndef RTX_BVH8
#define RTX_BVH8 0
#endif
#define RTX_LEAF8_MAX 3     // triangles per leaf child the masks have room for (3 bits per slot); the builder emits RTX_MAX_LEAF
struct Node8 {
    float ox, oy, oz;       // quantisation origin = node box min
    uint8_t ex, ey, ez, imask;   // biased exponents of the per-axis scale; imask bit s: slot s holds an inner node
    uint32_t childBase;     // node index of the first inner child
    uint32_t triBase;       // record index of the first triangle of the leaf children
    uint32_t tvalid;        // bit 3s+j: the leaf child in slot s has a j-th triangle; its record is triBase + popcount(tvalid below that bit)
                            // [31:24] tmask: slot s's subtree (or leaf) holds a surface that lets light through (no TRI_OPAQUE)
    uint32_t ttrans;        // bit 3s+j: that triangle lets light through; [31:24] valid: slot s holds a child at all
    uint8_t qlo[3][8];      // [axis][slot], box lo = o + q * scale (rounded down)
    uint8_t qhi[3][8];      // box hi (rounded up)
    uint32_t pad[12];
};
static_assert(sizeof(Node8) == 128, "Node8 must be 128 bytes");
#if RTX_BVH8
typedef Node8 BvhNode;
#else
typedef Node4 BvhNode;
#endif

struct TriRec {
    float v0x, v0y, v0z;
    uint32_t prim;          // global triangle id = faceBase[obj] + face (object order, then face order)
    float e1x, e1y, e1z;
    uint32_t kind;          // 0 = triangle in world space (identity node transform); RTX_KIND_XFTRI = triangle of an object
                            // under a non-identity node transform: v0, e1, e2 are the OBJECT-space operands and the ray
                            // is carried to node coordinates for the test (`obj` names the transform), as
                            // IntersectNodeAdapter.h:62,93 does; RTX_OBJ_SPHERE / RTX_OBJ_PLANE: analytic object, then
                            // v0 = world box min, e1 = e2 = box extent (so v0, v0+e1 span its box)
    float e2x, e2y, e2z;
    uint32_t obj;           // object index (read for analytic and transformed records only) | TRI_OPAQUE: the object's material lets no
                            // light through (Visible() returns 0 at it). Kept out of `kind`, which the triangle test reads.
                            // | TRI_INTERIOR: a world-space triangle that lies inside the root's box and its object's box by
                            // twice the margin of well_inside_box (traverse.h) for any hit distance below 1000: the reference's
                            // box tests cannot cull a hit on it, and the box-cull checks of the lean kernels need no box fetch
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");
#define RTX_KIND_XFTRI 3u
#define TRI_OBJ_MASK 0x3FFFFFFFu
#define TRI_OPAQUE 0x80000000u
#define TRI_INTERIOR 0x40000000u
#define TRI_INTERIOR_TMAX 1000.0f

// One 128-byte line per triangle holds everything the shading of a hit reads (surface.h resolve_triangle): the attributes and
// a copy of the TriRec words it needs (edges, kind, object word, primitive id). A hit used to fetch its TriRec (48 bytes that
// straddle a line one time in four) and its attributes (64 bytes) from two arrays - 2.25 lines of 128 bytes - for 100 bytes used.
struct TriAttr {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    uint32_t obj_mtl;       // obj | (mtlID << 16)
    float e1x, e1y, e1z;    // = TriRec (copied when the build ends: bvh_build.cpp)
    uint32_t kind;
    float e2x, e2y, e2z;
    uint32_t obj;
    uint32_t prim;
    uint32_t pad[7];
};
static_assert(sizeof(TriAttr) == 128, "TriAttr must be 128 bytes");

#define OBJ_HAS_NORMALS 1u
#define OBJ_HAS_TEXCOORDS 2u
#define OBJ_TRANSFORMED 8u   // the object's node transform is not the identity (ToNodeCoords / FromNodeCoords apply)
#define OBJ_XFMESH 16u       // ... and it is a mesh: its triangles stay in object space (RTX_KIND_XFTRI records)
struct ObjRec {
    int material;           // index into materials, -1 = none
    uint32_t flags;
    int face_base;          // first global triangle id of this object
    int nf;
};
// object <-> world transforms (Node::ToNodeCoords / FromNodeCoords), used by analytic objects and transformed meshes
struct ObjXform {
    float fwd[12];          // object -> world, 3x4 row-major
    float inv[12];          // world -> object
    float box[6];           // Object::GetBoundBox() in object coordinates: min xyz, max xyz (the box of IntersectNodeAdapter.h:63-75)
    float pad[2];
};

struct DevTexMap {
    const void* texels;     // uchar4 (RGB8) or float4 (RGB32F)
    int format, width, height, pad;
    float xf[12];
};

struct DevMaterial {
    int type;
    float kd[3]; int kd_tex;
    float ks[3]; int ks_tex;
    float kr[3]; int kr_tex;
    float kt[3]; int kt_tex;
    float ke[3]; int ke_tex;
    float gloss, ior;
    int gloss_tex;
    float absorption[3];
    int bump_tex;
    float bump_scale;
};

struct DevLight {
    float pos[3];
    float size;
    float intensity[3];
    float attenuation;
    // Terms of PointLight::GenerateSample / GetSampleInfo that depend on the light alone (Shading.cpp:154-160, 198-204), formed
    // once at scene creation by the same fp32 operations in the same order (rtx_api.cpp): radiance = intensity / (pi size^2),
    // radAtt = radiance / attenuation^2 (attenuation 0 counts as 1), pdfArea = 2 / (4 pi size^2). Sphere lights only (size > 0).
    float radiance[3];
    float pdfArea;
    float radAtt[3];
    float pad;
};

// Everything a kernel needs, passed by value as a kernel argument (lives in SGPRs / kernarg segment).
struct DevScene {
    const BvhNode* nodes;
    const TriRec* tris;
    const TriAttr* attrs;
    const ObjRec* objs;
    const ObjXform* objXf;
    const DevMaterial* mtls;
    const DevTexMap* texmaps;
    uint32_t root;          // child reference of the root (a leaf reference for tiny scenes)
    int n_tris, n_nodes, n_objs, n_lights, n_mtls, n_texmaps;
    float bounds[6];        // rootNode.GetChildBoundBox()
    DevLight lights[8];
    float env[3]; int env_tex;
    float bg[3];  int bg_tex;
    float bg_emission;
    int has_xf;             // some record is not a world-space triangle (a mesh under a non-identity node transform, a Sphere, a Plane): selects the kernel
    // what a function templated on the scene type may assume: DevSceneT<false> (traverse.h) overrides this with "world-space
    // triangles only"; a plain DevScene makes no such promise
    static constexpr bool kXF = true;
                            // instantiations that carry the object-space triangle test (traverse.h DevSceneT)
};
