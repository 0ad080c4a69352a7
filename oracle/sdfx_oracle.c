/*
 * sdfx_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A scalar, single-threaded C restatement of the algorithms behind the four native
 * extensions of ashawkey/stable-dreamfusion (raymarching, gridencoder, freqencoder,
 * shencoder).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The shipped HIP path never calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout).  Arithmetic follows the reference expression by expression,
 * including int->float / float->double promotions of the C++ source, and is compiled with
 * -ffp-contract=off so that no FMA is introduced that the source does not spell out.
 *
 * Parity pin status: the reference ships no tests or golden
 * vectors for this path.  The oracle is pinned against (a) analytic known answers,
 * (b) the reference's own Python code where an equivalent exists (FreqEncoder_torch,
 * NeRFRenderer.run compositing, the torch MLP), and (c) outputs of the reference's CUDA
 * sources compiled in place for gfx950 (oracle/_ref) and executed on the MI355X box,
 * committed as fixtures under tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------------------------ */
/* IEEE half <-> float, round-to-nearest-even (what at::Half / __half conversions do).   */
/* ------------------------------------------------------------------------------------ */
typedef uint16_t half_t;

static float h2f(half_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static half_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        return (half_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
    }
    if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (half_t)(sign | 0x7c00u);
    }
    if (ax < 0x33000001u) { /* <= 2^-25: rounds to zero (tie at exactly 2^-25 goes to even = 0) */
        return (half_t)sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    int shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = 13 + (-14 - e);
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t mant = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (mant & 1u))) mant++;
    uint32_t out;
    if (hexp == 0) {
        out = mant; /* may carry into exponent 1: correct by construction */
    } else {
        out = ((hexp - 1) << 10) + mant; /* mant includes the implicit bit (0x400) */
    }
    return (half_t)(sign | out);
}

/* exported for tests */
float orc_half_to_float(uint16_t h) { return h2f(h); }
uint16_t orc_float_to_half(float f) { return f2h(f); }

/* ------------------------------------------------------------------------------------ */
/* helpers: raymarching/src/raymarching.cu:19-81                                         */
/* ------------------------------------------------------------------------------------ */
#define ORC_SQRT3 1.7320508075688772f
#define ORC_RPI 0.3183098861837907f
#define ORC_PI 3.141592653589793f

static inline float signf_(float x) { return copysignf(1.0f, x); }                 /* :30-32 */
static inline float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); } /* :34-36 */

static inline int mip_from_pos(float x, float y, float z, float max_cascade) {      /* :42-47 */
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

static inline int mip_from_dt(float dt, float H, float max_cascade) {               /* :49-54 */
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

static inline uint32_t expand_bits(uint32_t v) {                                    /* :56-63 */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline uint32_t morton3D_(uint32_t x, uint32_t y, uint32_t z) {              /* :65-71 */
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

static inline uint32_t morton3D_invert_(uint32_t x) {                               /* :73-81 */
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* ------------------------------------------------------------------------------------ */
/* utils                                                                                */
/* ------------------------------------------------------------------------------------ */

/* raymarching.cu:91-145 */
void orc_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                            uint32_t N, float min_near, float* nears, float* fars) {
    for (uint32_t n = 0; n < N; n++) {
        const float* o = rays_o + (size_t)n * 3;
        const float* d = rays_d + (size_t)n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float dx = d[0], dy = d[1], dz = d[2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;

        float near = (aabb[0] - ox) * rdx;
        float far = (aabb[3] - ox) * rdx;
        if (near > far) { float c = near; near = far; far = c; }

        float near_y = (aabb[1] - oy) * rdy;
        float far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }

        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }

        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;

        float near_z = (aabb[2] - oz) * rdz;
        float far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }

        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }

        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;

        if (near < min_near) near = min_near;

        nears[n] = near;
        fars[n] = far;
    }
}

/* raymarching.cu:162-198 */
void orc_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    for (uint32_t n = 0; n < N; n++) {
        const float* o = rays_o + (size_t)n * 3;
        const float* d = rays_d + (size_t)n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float dx = d[0], dy = d[1], dz = d[2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float B = ox * dx + oy * dy + oz * dz;
        const float C = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-B + sqrtf(B * B - A * C)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        coords[(size_t)n * 2 + 0] = 2 * theta * ORC_RPI - 1;
        coords[(size_t)n * 2 + 1] = phi * ORC_RPI;
    }
}

/* raymarching.cu:214-226 */
void orc_morton3D(const int* coords, uint32_t N, int* indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int)morton3D_((uint32_t)coords[(size_t)n * 3], (uint32_t)coords[(size_t)n * 3 + 1],
                                    (uint32_t)coords[(size_t)n * 3 + 2]);
}

/* raymarching.cu:237-254 */
void orc_morton3D_invert(const int* indices, uint32_t N, int* coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int ind = indices[n];
        coords[(size_t)n * 3 + 0] = (int)morton3D_invert_((uint32_t)(ind >> 0));
        coords[(size_t)n * 3 + 1] = (int)morton3D_invert_((uint32_t)(ind >> 1));
        coords[(size_t)n * 3 + 2] = (int)morton3D_invert_((uint32_t)(ind >> 2));
    }
}

/* raymarching.cu:267-289; N = number of output bytes */
void orc_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        const float* g = grid + (size_t)n * 8;
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (g[i] > density_thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* raymarching.cu:303-319 */
void orc_flatten_rays(const int* rays, uint32_t N, uint32_t M, int* res) {
    (void)M;
    for (uint32_t n = 0; n < N; n++) {
        uint32_t offset = (uint32_t)rays[(size_t)n * 2];
        uint32_t num_steps = (uint32_t)rays[(size_t)n * 2 + 1];
        for (uint32_t i = 0; i < num_steps; i++) res[offset + i] = (int)n;
    }
}

/* ------------------------------------------------------------------------------------ */
/* one DDA step shared by the train / inference marchers                                 */
/* raymarching.cu:396-464 (train) and :760-827 (inference) are the same body.            */
/* Returns 1 when the sample at t is occupied (caller emits), else advances *t.          */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3;
    uint32_t C, H;
    int contract;
    const uint8_t* grid;
} march_ctx;

static inline int march_probe(const march_ctx* c, float* t_io, float* dt_out, float* cx_o, float* cy_o, float* cz_o) {
    float t = *t_io;
    const float bound = c->bound;
    const float x = clampf_(c->ox + t * c->dx, -bound, bound);
    const float y = clampf_(c->oy + t * c->dy, -bound, bound);
    const float z = clampf_(c->oz + t * c->dz, -bound, bound);

    float dt = clampf_(t * c->dt_gamma, c->dt_min, c->dt_max);

    const int la = mip_from_pos(x, y, z, (float)c->C);
    const int lb = mip_from_dt(dt, (float)c->H, (float)c->C);
    const int level = la > lb ? la : lb;

    const float mip_bound = fminf(scalbnf(1.0f, level), bound);
    const float mip_rbound = 1 / mip_bound;

    float cx = x, cy = y, cz = z;
    const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    if (c->contract && mag > 1) {
        const float Linf_scale = (2 - 1 / mag) / mag;
        cx *= Linf_scale;
        cy *= Linf_scale;
        cz *= Linf_scale;
    }

    /* `0.5 * (cx * mip_rbound + 1) * H` is evaluated in double in the reference (0.5 is a double literal) */
    const float Hm1 = (float)(c->H - 1);
    const int nx = (int)clampf_((float)(0.5 * (double)(cx * mip_rbound + 1) * (double)c->H), 0.0f, Hm1);
    const int ny = (int)clampf_((float)(0.5 * (double)(cy * mip_rbound + 1) * (double)c->H), 0.0f, Hm1);
    const int nz = (int)clampf_((float)(0.5 * (double)(cz * mip_rbound + 1) * (double)c->H), 0.0f, Hm1);

    /* index arithmetic in float, exactly as the reference (H3 is a float) */
    const uint32_t index = (uint32_t)((float)level * c->H3 + (float)morton3D_((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = c->grid[index / 8] & (1 << (index % 8));

    *dt_out = dt;
    *cx_o = cx; *cy_o = cy; *cz_o = cz;

    if (occ) return 1;

    if (c->contract && mag > 1) {
        t += dt;
    } else {
        const float tx = (((nx + 0.5f + 0.5f * signf_(c->dx)) * c->rH * 2 - 1) * mip_bound - cx) * c->rdx;
        const float ty = (((ny + 0.5f + 0.5f * signf_(c->dy)) * c->rH * 2 - 1) * mip_bound - cy) * c->rdy;
        const float tz = (((nz + 0.5f + 0.5f * signf_(c->dz)) * c->rH * 2 - 1) * mip_bound - cz) * c->rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do {
            dt = clampf_(t * c->dt_gamma, c->dt_min, c->dt_max);
            t += dt;
        } while (t < tt);
    }
    *t_io = t;
    return 0;
}

static void march_ctx_init(march_ctx* c, const float* o, const float* d, const uint8_t* grid, float bound, int contract,
                           float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2];
    c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    c->rdx = 1 / c->dx; c->rdy = 1 / c->dy; c->rdz = 1 / c->dz;
    c->rH = 1 / (float)H;
    c->H3 = (float)(H * H * H);
    c->bound = bound;
    c->dt_gamma = dt_gamma;
    c->dt_min = 2 * ORC_SQRT3 / max_steps;
    c->dt_max = 2 * ORC_SQRT3 * bound / H;
    c->C = C; c->H = H;
    c->contract = contract;
    c->grid = grid;
}

/*
 * raymarching.cu:337-475.  xyzs == NULL selects the first (counting) pass.  The reference
 * hands out offsets with atomicAdd in whatever order threads finish; a serial loop in ray
 * order gives the exclusive prefix sum of the counts, which is the deterministic order the
 * HIP path also produces.
 */
void orc_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                          const float* nears, const float* fars, float* xyzs, float* dirs, float* ts,
                          int* rays, int* counter, const float* noises) {
    const int first_pass = (xyzs == NULL);
    for (uint32_t n = 0; n < N; n++) {
        march_ctx c;
        march_ctx_init(&c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, bound, contract, dt_gamma, max_steps, C, H);
        int* ray = rays + (size_t)n * 2;
        uint32_t num_steps = max_steps;
        float *px = NULL, *pd = NULL, *pt = NULL;
        if (!first_pass) {
            uint32_t point_index = (uint32_t)ray[0];
            num_steps = (uint32_t)ray[1];
            px = xyzs + (size_t)point_index * 3;
            pd = dirs + (size_t)point_index * 3;
            pt = ts + (size_t)point_index * 2;
        }
        const float near = nears[n], far = fars[n], noise = noises[n];
        float t0 = near;
        t0 += clampf_(t0 * dt_gamma, c.dt_min, c.dt_max) * noise;
        float t = t0;
        uint32_t step = 0;
        while (t < far && step < num_steps) {
            float dt, cx, cy, cz;
            if (march_probe(&c, &t, &dt, &cx, &cy, &cz)) {
                step++;
                t += dt;
                if (!first_pass) {
                    px[0] = cx; px[1] = cy; px[2] = cz;
                    pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                    pt[0] = t; pt[1] = dt;
                    px += 3; pd += 3; pt += 2;
                }
            }
        }
        if (first_pass) {
            uint32_t point_index = (uint32_t)counter[0];
            counter[0] += (int)step;
            ray[0] = (int)point_index;
            ray[1] = (int)step;
        }
    }
}

/* raymarching.cu:500-579 */
void orc_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ts, const int* rays,
                                      uint32_t M, uint32_t N, float T_thresh, int binarize,
                                      float* weights, float* weights_sum, float* depth, float* image) {
    for (uint32_t n = 0; n < N; n++) {
        uint32_t offset = (uint32_t)rays[(size_t)n * 2];
        uint32_t num_steps = (uint32_t)rays[(size_t)n * 2 + 1];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[n] = 0; depth[n] = 0;
            image[(size_t)n * 3] = 0; image[(size_t)n * 3 + 1] = 0; image[(size_t)n * 3 + 2] = 0;
            continue;
        }
        const float* pts = ts + (size_t)offset * 2;
        float* pw = weights + offset;
        const float* ps = sigmas + offset;
        const float* pc = rgbs + (size_t)offset * 3;
        uint32_t step = 0;
        float T = 1.0f;
        float r = 0, g = 0, b = 0, ws = 0, d = 0;
        while (step < num_steps) {
            const float real_alpha = 1.0f - expf(-ps[0] * pts[1]);
            const float alpha = binarize ? (real_alpha > 0.5 ? 1.0f : 0.0f) : real_alpha;
            const float weight = alpha * T;
            pw[0] = weight;
            r += weight * pc[0];
            g += weight * pc[1];
            b += weight * pc[2];
            ws += weight;
            d += weight * pts[0];
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            pw++; ps++; pc += 3; pts += 2;
            step++;
        }
        weights_sum[n] = ws;
        depth[n] = d;
        image[(size_t)n * 3] = r; image[(size_t)n * 3 + 1] = g; image[(size_t)n * 3 + 2] = b;
    }
}

/* raymarching.cu:605-695 */
void orc_composite_rays_train_backward(const float* grad_weights, const float* grad_weights_sum, const float* grad_depth,
                                       const float* grad_image, const float* sigmas, const float* rgbs, const float* ts,
                                       const int* rays, const float* weights_sum, const float* depth, const float* image,
                                       uint32_t M, uint32_t N, float T_thresh, int binarize,
                                       float* grad_sigmas, float* grad_rgbs) {
    for (uint32_t n = 0; n < N; n++) {
        uint32_t offset = (uint32_t)rays[(size_t)n * 2];
        uint32_t num_steps = (uint32_t)rays[(size_t)n * 2 + 1];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float* gw = grad_weights + offset;
        const float gws = grad_weights_sum[n];
        const float gd = grad_depth[n];
        const float* gi = grad_image + (size_t)n * 3;
        const float* ps = sigmas + offset;
        const float* pc = rgbs + (size_t)offset * 3;
        const float* pts = ts + (size_t)offset * 2;
        float* gs = grad_sigmas + offset;
        float* gc = grad_rgbs + (size_t)offset * 3;
        uint32_t step = 0;
        float T = 1.0f;
        const float r_final = image[(size_t)n * 3], g_final = image[(size_t)n * 3 + 1], b_final = image[(size_t)n * 3 + 2];
        const float ws_final = weights_sum[n], d_final = depth[n];
        float r = 0, g = 0, b = 0, ws = 0, d = 0;
        while (step < num_steps) {
            const float real_alpha = 1.0f - expf(-ps[0] * pts[1]);
            const float alpha = binarize ? (real_alpha > 0.5 ? 1.0f : 0.0f) : real_alpha;
            const float weight = alpha * T;
            r += weight * pc[0];
            g += weight * pc[1];
            b += weight * pc[2];
            ws += weight;
            d += weight * pts[0];
            T *= 1.0f - alpha;
            gc[0] = gi[0] * weight;
            gc[1] = gi[1] * weight;
            gc[2] = gi[2] * weight;
            gs[0] = pts[1] * (
                gi[0] * (T * pc[0] - (r_final - r)) +
                gi[1] * (T * pc[1] - (g_final - g)) +
                gi[2] * (T * pc[2] - (b_final - b)) +
                (gws + gw[0]) * (T - (ws_final - ws)) +
                gd * (T * pts[0] - (d_final - d)));
            if (T < T_thresh) break;
            ps++; pc += 3; pts += 2; gw++; gs++; gc += 3;
            step++;
        }
    }
}

/* raymarching.cu:713-829 */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int* rays_alive, const float* rays_t,
                    const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                    uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                    const float* fars, float* xyzs, float* dirs, float* ts, const float* noises) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        const float noise = noises[n];
        march_ctx c;
        march_ctx_init(&c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid, bound, contract, dt_gamma, max_steps, C, H);
        float* px = xyzs + (size_t)n * n_step * 3;
        float* pd = dirs + (size_t)n * n_step * 3;
        float* pt = ts + (size_t)n * n_step * 2;
        const float far = fars[index];
        (void)nears;
        float t = rays_t[index];
        t += clampf_(t * dt_gamma, c.dt_min, c.dt_max) * noise;
        uint32_t step = 0;
        while (t < far && step < n_step) {
            float dt, cx, cy, cz;
            if (march_probe(&c, &t, &dt, &cx, &cy, &cz)) {
                px[0] = cx; px[1] = cy; px[2] = cz;
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                t += dt;
                pt[0] = t; pt[1] = dt;
                px += 3; pd += 3; pt += 2;
                step++;
            }
        }
    }
}

/* raymarching.cu:842-925 */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int binarize, int* rays_alive, float* rays_t,
                        const float* sigmas, const float* rgbs, const float* ts, float* weights_sum, float* depth,
                        float* image) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        const float* ps = sigmas + (size_t)n * n_step;
        const float* pc = rgbs + (size_t)n * n_step * 3;
        const float* pts = ts + (size_t)n * n_step * 2;
        float t = 0; /* reference leaves it uninitialised; only read when step == n_step >= 1 */
        float d = depth[index], r = image[(size_t)index * 3], g = image[(size_t)index * 3 + 1], b = image[(size_t)index * 3 + 2];
        float weight_sum = weights_sum[index];
        uint32_t step = 0;
        while (step < n_step) {
            if (pts[0] == 0) break;
            const float real_alpha = 1.0f - expf(-ps[0] * pts[1]);
            const float alpha = binarize ? (real_alpha > 0.5 ? 1.0f : 0.0f) : real_alpha;
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t = pts[0];
            d += weight * t;
            r += weight * pc[0];
            g += weight * pc[1];
            b += weight * pc[2];
            if (T < T_thresh) break;
            ps++; pc += 3; pts += 2;
            step++;
        }
        if (step < n_step) rays_alive[n] = -1;
        else rays_t[index] = t;
        weights_sum[index] = weight_sum;
        depth[index] = d;
        image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
    }
}

/* ------------------------------------------------------------------------------------ */
/* gridencoder                                                                          */
/* ------------------------------------------------------------------------------------ */
#define ORC_MAX_D 5
#define ORC_MAX_C 32

/* table element access: the reference templates on scalar_t in {float, at::Half} */
static inline float tab_ld(const void* tab, size_t i, int is_half) {
    return is_half ? h2f(((const half_t*)tab)[i]) : ((const float*)tab)[i];
}
static inline void tab_st(void* tab, size_t i, float v, int is_half) {
    if (is_half) ((half_t*)tab)[i] = f2h(v); else ((float*)tab)[i] = v;
}
/* value as scalar_t would hold it */
static inline float as_scalar(float v, int is_half) { return is_half ? h2f(f2h(v)) : v; }

/* gridencoder.cu:45-58 */
static inline uint32_t fast_hash(const uint32_t* pos_grid, uint32_t D) {
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
    return result;
}

/* gridencoder.cu:61-79 */
static inline uint32_t get_grid_index(uint32_t gridtype, uint32_t D, uint32_t C, uint32_t ch, uint32_t hashmap_size,
                                      uint32_t resolution, const uint32_t* pos_grid) {
    uint32_t stride = 1;
    uint32_t index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= resolution;
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash(pos_grid, D);
    return (index % hashmap_size) * C + ch;
}

/* gridencoder.cu:133 — (uint32_t)ceil(exp2f(level * S) * H), float arithmetic throughout */
uint32_t orc_grid_resolution(uint32_t level, float S, uint32_t H) {
    return (uint32_t)ceilf(exp2f((float)level * S) * (float)H);
}

static inline float smoothstep_(float v) { return v * v * (3.0f - 2.0f * v); }          /* :34-37 */
static inline float smoothstep_deriv_(float v) { return 6 * v * (1.0f - v); }           /* :39-42 */

/* position → (pos frac, pos_grid) : gridencoder.cu:140-160 */
static inline void grid_locate(const float* in, uint32_t D, uint32_t resolution, int align_corners, uint32_t interp,
                               float* pos, float* pos_deriv, uint32_t* pos_grid) {
    for (uint32_t d = 0; d < D; d++) {
        if (align_corners) {
            pos[d] = in[d] * (float)(resolution - 1);
            uint32_t f = (uint32_t)floorf(pos[d]);
            pos_grid[d] = f < resolution - 2 ? f : resolution - 2;
        } else {
            pos[d] = fminf(fmaxf(in[d] * (float)resolution - 0.5f, 0.0f), (float)(resolution - 1));
            pos_grid[d] = (uint32_t)floorf(pos[d]);
        }
        pos[d] -= (float)pos_grid[d];
        if (interp == 1) {
            if (pos_deriv) pos_deriv[d] = smoothstep_deriv_(pos[d]);
            pos[d] = smoothstep_(pos[d]);
        } else {
            if (pos_deriv) pos_deriv[d] = 1.0f;
        }
    }
}

/*
 * gridencoder.cu:82-249 (kernel_grid) — outputs [L, B, C]; dy_dx [B, L, D, C] or NULL.
 * is_half selects scalar_t = at::Half: the 8-corner sum is then accumulated in half
 * (`results[ch] += w * grid[...]` with results of type scalar_t, :168,191): each product is rounded to half
 * (implicit float -> at::Half conversion of the right-hand side) and each partial sum is rounded to half.
 * Verified bit for bit against the reference kernel itself (tests/test_gpu_vs_reference_kernels.py).
 */
void orc_grid_encode_forward(const float* inputs, const void* embeddings, const int* offsets, void* outputs,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                             void* dy_dx, uint32_t gridtype, int align_corners, uint32_t interp, int is_half) {
    for (uint32_t level = 0; level < max_level; level++) {
        const size_t tab0 = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t resolution = orc_grid_resolution(level, S, H);
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (size_t)b * D;
            const size_t out0 = (size_t)level * B * C + (size_t)b * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) tab_st(outputs, out0 + ch, 0.0f, is_half);
                if (dy_dx) {
                    const size_t dy0 = (size_t)b * D * L * C + (size_t)level * D * C;
                    for (uint32_t i = 0; i < D * C; i++) tab_st(dy_dx, dy0 + i, 0.0f, is_half);
                }
                continue;
            }
            float pos[ORC_MAX_D], pos_deriv[ORC_MAX_D];
            uint32_t pos_grid[ORC_MAX_D];
            grid_locate(in, D, resolution, align_corners, interp, pos, pos_deriv, pos_grid);

            float results[ORC_MAX_C];
            for (uint32_t ch = 0; ch < C; ch++) results[ch] = 0;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) {
                        w *= 1 - pos[d];
                        pgl[d] = pos_grid[d];
                    } else {
                        w *= pos[d];
                        pgl[d] = pos_grid[d] + 1 < resolution - 1 ? pos_grid[d] + 1 : resolution - 1;
                    }
                }
                const uint32_t index = get_grid_index(gridtype, D, C, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++)
                    /* `results[ch] += w * grid[...]` with results of type at::Half resolves to operator+=(Half&, const Half&):
                       the float product is converted (rounded) to Half first, then the two halves are added and rounded */
                    results[ch] = as_scalar(results[ch] + as_scalar(w * tab_ld(embeddings, tab0 + index + ch, is_half), is_half), is_half);
            }
            for (uint32_t ch = 0; ch < C; ch++) tab_st(outputs, out0 + ch, results[ch], is_half);

            if (dy_dx) {
                const size_t dy0 = (size_t)b * D * L * C + (size_t)level * D * C;
                for (uint32_t gd = 0; gd < D; gd++) {
                    float rg[ORC_MAX_C];
                    for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0;
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = (float)(align_corners ? resolution - 1 : resolution);
                        uint32_t pgl[ORC_MAX_D];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) {
                                w *= 1 - pos[d];
                                pgl[d] = pos_grid[d];
                            } else {
                                w *= pos[d];
                                pgl[d] = pos_grid[d] + 1 < resolution - 1 ? pos_grid[d] + 1 : resolution - 1;
                            }
                        }
                        pgl[gd] = pos_grid[gd];
                        const uint32_t il = get_grid_index(gridtype, D, C, 0, hashmap_size, resolution, pgl);
                        pgl[gd] = pos_grid[gd] + 1 < resolution - 1 ? pos_grid[gd] + 1 : resolution - 1;
                        const uint32_t ir = get_grid_index(gridtype, D, C, 0, hashmap_size, resolution, pgl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            /* (grid[r] - grid[l]) is a scalar_t subtraction (:239) */
                            const float diff = as_scalar(tab_ld(embeddings, tab0 + ir + ch, is_half) - tab_ld(embeddings, tab0 + il + ch, is_half), is_half);
                            rg[ch] = as_scalar(rg[ch] + as_scalar(w * diff * pos_deriv[gd], is_half), is_half);  /* same double rounding */
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) tab_st(dy_dx, dy0 + (size_t)gd * C + ch, rg[ch], is_half);
                }
            }
        }
    }
}

/*
 * gridencoder.cu:252-349 (kernel_grid_backward) + :352-378 (kernel_input_backward).
 * grad [L, B, C]; grad_embeddings pre-zeroed by the caller; serial accumulation in
 * (level, b, corner) order — the GPU order is whatever the atomics give.
 * half path: each contribution is rounded to half, then added in half (:338-339).
 */
void orc_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int* offsets,
                              void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                              float S, uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                              int align_corners, uint32_t interp, int is_half) {
    (void)embeddings;
    for (uint32_t level = 0; level < max_level; level++) {
        const size_t tab0 = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t resolution = orc_grid_resolution(level, S, H);
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;
            float pos[ORC_MAX_D];
            uint32_t pos_grid[ORC_MAX_D];
            grid_locate(in, D, resolution, align_corners, interp, pos, NULL, pos_grid);
            const size_t g0 = (size_t)level * B * C + (size_t)b * C;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) {
                        w *= 1 - pos[d];
                        pgl[d] = pos_grid[d];
                    } else {
                        w *= pos[d];
                        pgl[d] = pos_grid[d] + 1 < resolution - 1 ? pos_grid[d] + 1 : resolution - 1;
                    }
                }
                const uint32_t index = get_grid_index(gridtype, D, C, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float gcur = tab_ld(grad, g0 + ch, is_half);
                    const float contrib = as_scalar(w * gcur, is_half);
                    const float old = tab_ld(grad_embeddings, tab0 + index + ch, is_half);
                    tab_st(grad_embeddings, tab0 + index + ch, old + contrib, is_half);
                }
            }
        }
    }
    if (dy_dx && grad_inputs) {
        for (uint32_t t = 0; t < B * D; t++) {
            const uint32_t b = t / D, d = t - b * D;
            const size_t dy0 = (size_t)b * L * D * C;
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++) {
                    /* scalar_t product then scalar_t accumulate (:373) */
                    const float prod = as_scalar(tab_ld(grad, (size_t)l * B * C + (size_t)b * C + ch, is_half) *
                                                 tab_ld(dy_dx, dy0 + (size_t)l * D * C + (size_t)d * C + ch, is_half), is_half);
                    result = as_scalar(result + prod, is_half);
                }
            tab_st(grad_inputs, t, result, is_half);
        }
    }
}

/* gridencoder.cu:525-631 (kernel_grad_tv), float tables only (always called with autocast off, grid.py:172) */
void orc_grad_total_variation(const float* inputs, const float* embeddings, float* grad, const int* offsets, float weight,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                              int align_corners) {
    for (uint32_t level = 0; level < L; level++) {
        const size_t tab0 = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t resolution = orc_grid_resolution(level, S, H);
        for (uint32_t b = 0; b < B; b++) {
            const float* in = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (in[d] < 0 || in[d] > 1) oob = 1;
            if (oob) continue;
            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                float p;
                if (align_corners) {
                    p = in[d] * (float)(resolution - 1);
                    uint32_t f = (uint32_t)floorf(p);
                    pos_grid[d] = f < resolution - 2 ? f : resolution - 2;
                } else {
                    p = fminf(fmaxf(in[d] * (float)resolution - 0.5f, 0.0f), (float)(resolution - 1));
                    pos_grid[d] = (uint32_t)floorf(p);
                }
            }
            float results[ORC_MAX_C], idelta[ORC_MAX_C];
            for (uint32_t ch = 0; ch < C; ch++) { results[ch] = 0; idelta[ch] = 0; }
            const uint32_t index = get_grid_index(gridtype, D, C, 0, hashmap_size, resolution, pos_grid);
            const float w = weight / (2 * D);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur_d = pos_grid[d];
                if (cur_d < resolution) {
                    pos_grid[d] = cur_d + 1;
                    const uint32_t ir = get_grid_index(gridtype, D, C, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = embeddings[tab0 + index + ch] - embeddings[tab0 + ir + ch];
                        results[ch] += gv;
                        idelta[ch] += gv * gv;
                    }
                }
                if (cur_d > 0) {
                    pos_grid[d] = cur_d - 1;
                    const uint32_t il = get_grid_index(gridtype, D, C, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = embeddings[tab0 + index + ch] - embeddings[tab0 + il + ch];
                        results[ch] += gv;
                        idelta[ch] += gv * gv;
                    }
                }
                pos_grid[d] = cur_d;
            }
            for (uint32_t ch = 0; ch < C; ch++)
                grad[tab0 + index + ch] += w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f));
        }
    }
}

/* gridencoder.cu:670-703 (kernel_grad_wd); B = number of table rows */
void orc_grad_weight_decay(const float* embeddings, float* grad, const int* offsets, float weight, uint32_t B, uint32_t C,
                           uint32_t L) {
    for (uint32_t b = 0; b < B * C; b++) {
        uint32_t level = 0;
        const uint32_t n = b / C;
        uint32_t l = 0, r = L;
        while (l < r) {
            uint32_t m = (l + r) / 2;
            if ((uint32_t)offsets[m] <= n) { level = m; l = m + 1; } else { r = m; }
        }
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        grad[b] += 2 * weight * embeddings[b] / hashmap_size;
    }
}

/* ------------------------------------------------------------------------------------ */
/* freqencoder: freqencoder.cu:30-58 / 63-94                                             */
/* ------------------------------------------------------------------------------------ */
void orc_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    (void)deg;
    for (uint32_t t = 0; t < B * C; t++) {
        const uint32_t b = t / C, c = t - b * C;
        const float* in = inputs + (size_t)b * D;
        if (c < D) {
            outputs[t] = in[c];
        } else {
            const uint32_t col = c / D - 1;
            const uint32_t d = c % D;
            const uint32_t freq = col / 2;
            const float phase_shift = (col % 2) * (ORC_PI / 2);
            outputs[t] = sinf(scalbnf(in[d], (int)freq) + phase_shift);
        }
    }
}

void orc_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                              float* grad_inputs) {
    for (uint32_t t = 0; t < B * D; t++) {
        const uint32_t b = t / D, d = t - b * D;
        const float* g = grad + (size_t)b * C;
        const float* o = outputs + (size_t)b * C;
        float result = g[d];
        g += D; o += D;
        for (uint32_t f = 0; f < deg; f++) {
            result += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
            g += 2 * D; o += 2 * D;
        }
        grad_inputs[t] = result;
    }
}

/* ------------------------------------------------------------------------------------ */
/* shencoder: shencoder.cu:27-355 (kernel_sh) / 358-382 (kernel_sh_backward)             */
/*                                                                                      */
/* The reference spells out 64 closed-form polynomials (+192 derivatives).  They are the */
/* real spherical harmonics written as polynomials of (x, y, z):                         */
/*   Y_l^m = (-1)^m sqrt2 K_l^|m| * d^|m|P_l/dz^|m| (z) * {Re | Im}((x + i y)^|m|)       */
/* with K_l^m = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!), Re for m > 0, Im for m < 0, and the   */
/* Legendre part kept as a polynomial in z alone (x²+y² is NOT folded in).  The oracle   */
/* evaluates that definition in double precision with recurrences, then rounds to float; */
/* tests/golden/sh_ref.npz pins it to the reference's literal expressions.               */
/* ------------------------------------------------------------------------------------ */
#define ORC_SH_MAXDEG 8

/* derivatives of Legendre polynomials: P[l][k] = coefficient of z^k of P_l(z) */
static void legendre_coeffs(double P[ORC_SH_MAXDEG][ORC_SH_MAXDEG]) {
    memset(P, 0, sizeof(double) * ORC_SH_MAXDEG * ORC_SH_MAXDEG);
    P[0][0] = 1.0;
    if (ORC_SH_MAXDEG > 1) P[1][1] = 1.0;
    for (int l = 2; l < ORC_SH_MAXDEG; l++) {
        /* l P_l = (2l-1) z P_{l-1} - (l-1) P_{l-2} */
        for (int k = 0; k < ORC_SH_MAXDEG; k++) {
            double a = (k > 0) ? (2.0 * l - 1.0) * P[l - 1][k - 1] : 0.0;
            double b = (l - 1.0) * P[l - 2][k];
            P[l][k] = (a - b) / l;
        }
    }
}

static double factorial_(int n) { double f = 1; for (int i = 2; i <= n; i++) f *= i; return f; }

/* value and z-derivative of d^m P_l / dz^m at z */
static void legendre_dm(const double P[ORC_SH_MAXDEG][ORC_SH_MAXDEG], int l, int m, double z, double* q, double* dq) {
    double c[ORC_SH_MAXDEG];
    for (int k = 0; k < ORC_SH_MAXDEG; k++) c[k] = P[l][k];
    for (int i = 0; i < m; i++) { /* differentiate m times */
        for (int k = 0; k + 1 < ORC_SH_MAXDEG; k++) c[k] = c[k + 1] * (k + 1);
        c[ORC_SH_MAXDEG - 1] = 0;
    }
    double v = 0, dv = 0;
    for (int k = ORC_SH_MAXDEG - 1; k >= 0; k--) v = v * z + c[k];
    for (int k = ORC_SH_MAXDEG - 1; k >= 1; k--) dv = dv * z + c[k] * k;
    *q = v; *dq = dv;
}

/* fills y[deg*deg] and optionally dy[3][deg*deg] in double */
static void sh_eval(int deg, double x, double y, double z, double* out, double* dx, double* dy, double* dz) {
    double P[ORC_SH_MAXDEG][ORC_SH_MAXDEG];
    legendre_coeffs(P);
    /* A_m = Re (x+iy)^m, B_m = Im (x+iy)^m */
    double A[ORC_SH_MAXDEG], Bm[ORC_SH_MAXDEG];
    A[0] = 1; Bm[0] = 0;
    for (int m = 1; m < ORC_SH_MAXDEG; m++) {
        A[m] = A[m - 1] * x - Bm[m - 1] * y;
        Bm[m] = A[m - 1] * y + Bm[m - 1] * x;
    }
    const double pi = 3.14159265358979323846;
    for (int l = 0; l < deg; l++) {
        for (int m = -l; m <= l; m++) {
            const int am = m < 0 ? -m : m;
            const int idx = l * l + l + m;
            double K = sqrt((2.0 * l + 1.0) / (4.0 * pi) * factorial_(l - am) / factorial_(l + am));
            if (am > 0) K *= sqrt(2.0) * ((am & 1) ? -1.0 : 1.0);
            double q, dq;
            legendre_dm(P, l, am, z, &q, &dq);
            double ang, ang_dx, ang_dy;
            if (m > 0) {
                ang = A[am]; ang_dx = am * A[am - 1]; ang_dy = -am * Bm[am - 1];
            } else if (m < 0) {
                ang = Bm[am]; ang_dx = am * Bm[am - 1]; ang_dy = am * A[am - 1];
            } else {
                ang = 1; ang_dx = 0; ang_dy = 0;
            }
            out[idx] = K * q * ang;
            if (dx) {
                dx[idx] = K * q * ang_dx;
                dy[idx] = K * q * ang_dy;
                dz[idx] = K * dq * ang;
            }
        }
    }
}

/* inputs [B, 3]; outputs [B, C*C]; dy_dx [B, 3, C*C] or NULL (shencoder.cu:27-355) */
void orc_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx) {
    const uint32_t C2 = C * C;
    double o[64], gx[64], gy[64], gz[64];
    for (uint32_t b = 0; b < B; b++) {
        const float* in = inputs + (size_t)b * D;
        sh_eval((int)C, in[0], in[1], in[2], o, dy_dx ? gx : NULL, gy, gz);
        for (uint32_t i = 0; i < C2; i++) outputs[(size_t)b * C2 + i] = (float)o[i];
        if (dy_dx) {
            float* dd = dy_dx + (size_t)b * D * C2;
            for (uint32_t i = 0; i < C2; i++) {
                dd[i] = (float)gx[i];
                dd[C2 + i] = (float)gy[i];
                dd[2 * C2 + i] = (float)gz[i];
            }
        }
    }
}

/* shencoder.cu:358-382; grad_inputs is accumulated into (pre-zeroed by the caller) */
void orc_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx,
                            float* grad_inputs) {
    (void)inputs;
    const uint32_t C2 = C * C;
    for (uint32_t t = 0; t < B * D; t++) {
        const uint32_t b = t / D, d = t - b * D;
        const float* g = grad + (size_t)b * C2;
        const float* dd = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
        for (uint32_t ch = 0; ch < C2; ch++) grad_inputs[t] += g[ch] * dd[ch];
    }
}
