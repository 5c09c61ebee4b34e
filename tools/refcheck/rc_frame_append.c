/* tools/refcheck/rc_frame_append.c -- csrc/shim/frame-hip.c's hip_append against the reference's own bitstream code (src/bitstream.c):
 * bytes produced by uvg_bitstream_put_byte (emulation prevention applied, as the device's row coder leaves them) appended a chunk at a time
 * must leave the stream exactly as a uvg_bitstream_writebyte per byte does -- chunks, len -- and with the zerocount put_byte had left.
 * The shim is compiled as it is (included below); the library's and the rate control's entry points it names are stand-ins that are never
 * called.  TEST INFRASTRUCTURE (tests/test_frame_shim_host.py builds and runs it where /root/reference exists).  Prints "ok <cases>". */
#include "../../uvg266_amd/csrc/shim/frame-hip.c"

const char *uvghip_last_error(void) { return ""; }
int uvghip_init(int d) { (void)d; return 1; }
int uvghip_frame_pool_create_tiles(int a, const uvghip_ctu_params_t *b, int c, int d, int e, const int32_t *f, int g, const int32_t *h, int i, uvghip_frame_pool_t **j)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; (void)j; return 1; }
int uvghip_frame_pool_begin(uvghip_frame_pool_t *a, int b, const uvghip_ctu_params_t *c, const void *d, const void *e, const void *f, int g, int h)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; return 1; }
int uvghip_frame_pool_finish(uvghip_frame_pool_t *a, int b, void *c, void *d, void *e, int f, int g, const uint8_t **h, const int32_t **i, int *j)
{ (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i; (void)j; return 1; }
void uvghip_frame_pool_destroy(uvghip_frame_pool_t *a) { (void)a; }
void uvg_set_lcu_lambda_and_qp(encoder_state_t *const s, vector2d_t p) { (void)s; (void)p; }
double uvg_calculate_chroma_lambda(encoder_state_t *s, bool a, int b) { (void)s; (void)a; (void)b; return 0; }

static uint32_t rng_state = 12345;
static uint32_t rng(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static int flatten(const bitstream_t *s, uint8_t *out)
{
  int n = 0;
  for (const uvg_data_chunk *c = s->first; c; c = c->next) { memcpy(out + n, c->data, c->len); n += c->len; }
  return n;
}

int main(void)
{
  enum { MAXN = 20000 };
  static uint8_t a_bytes[2 * MAXN], b_bytes[2 * MAXN];
  int cases = 0;
  for (int t = 0; t < 400; ++t) {
    const int raw = t < 8 ? t : (int)(rng() % MAXN);                 /* 0 .. a few chunks; lengths around the 4096-byte chunk edges too */
    const int n_raw = t % 7 == 0 ? 4096 * (1 + t % 3) + (t % 5) - 2 : raw;
    bitstream_t a, b;
    uvg_bitstream_init(&a); uvg_bitstream_init(&b);
    for (int i = 0; i < (n_raw > 0 ? n_raw : 0); ++i) {
      const uint32_t r = rng();
      uvg_bitstream_put_byte(&a, (r & 3) ? 0 : (r >> 4) & (t & 1 ? 0xff : 3));   /* many zeros and small values: emulation prevention everywhere */
    }
    const int n = flatten(&a, a_bytes);
    /* appended in one, two or three pieces (a row's bytes arrive whole; a frame's rows one after the other into DIFFERENT streams, but the
     * function must not care) */
    const int cut1 = n ? (int)(rng() % (n + 1)) : 0, cut2 = cut1 + (n - cut1 ? (int)(rng() % (n - cut1 + 1)) : 0);
    if (t % 3 == 0) hip_append(&b, a_bytes, n);
    else { hip_append(&b, a_bytes, cut1); hip_append(&b, a_bytes + cut1, cut2 - cut1); hip_append(&b, a_bytes + cut2, n - cut2); }
    const int m = flatten(&b, b_bytes);
    if (m != n || (int)b.len != n || memcmp(a_bytes, b_bytes, n) || b.zerocount != a.zerocount || b.cur_bit != 0) {
      printf("case %d: n %d m %d len %u zerocount %d / %d\n", t, n, m, b.len, b.zerocount, a.zerocount);
      return 1;
    }
    for (const uvg_data_chunk *c = b.first; c; c = c->next) if (c->next && c->len != UVG_DATA_CHUNK_SIZE) { printf("case %d: a chunk that is not full inside the list\n", t); return 1; }
    uvg_bitstream_finalize(&a); uvg_bitstream_finalize(&b);
    ++cases;
  }
  printf("ok %d\n", cases);
  return 0;
}
