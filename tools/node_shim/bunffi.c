/* tools/node_shim/bunffi.c — TEST INFRASTRUCTURE: a minimal N-API stand-in for the parts of `bun:ffi` that
 * ts/spiceyHip.ts uses (dlopen / symbol calls / ptr / CString), so that the TypeScript drop-in layer (type-erased) can
 * be EXECUTED under the Node 12 of this image against libspicey_hip.so.  Bun itself is not available offline.
 *
 * call(): x86-64 SysV passes integer-class and double arguments in separate register files, in order of appearance
 * within each class, so one trampoline with 8 integer and 4 double parameters reaches every entry point of
 * include/spicey_hip.h (at most 7 pointer/integer + 1 double argument). */
#define NAPI_VERSION 6
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <string.h>

#define CHECK(x) do { if ((x) != napi_ok) { napi_throw_error(env, NULL, "bunffi: " #x); return NULL; } } while (0)

static int64_t to_i64(napi_env env, napi_value v) {
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (t == napi_bigint) { int64_t x = 0; bool lossless; uint64_t u; if (napi_get_value_bigint_int64(env, v, &x, &lossless) != napi_ok) x = 0; if (!lossless && napi_get_value_bigint_uint64(env, v, &u, &lossless) == napi_ok) x = (int64_t)u; return x; }
  if (t == napi_number) { double d = 0; napi_get_value_double(env, v, &d); return (int64_t)d; }
  if (t == napi_null || t == napi_undefined) return 0;
  if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); return b; }
  return 0;
}

static napi_value fn_dlopen(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  char path[4096]; size_t n = 0;
  CHECK(napi_get_value_string_utf8(env, argv[0], path, sizeof path, &n));
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { napi_throw_error(env, NULL, dlerror()); return NULL; }
  napi_value out; CHECK(napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)h, &out));
  return out;
}

static napi_value fn_dlsym(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  char name[256]; size_t n = 0;
  CHECK(napi_get_value_string_utf8(env, argv[1], name, sizeof name, &n));
  void *p = dlsym((void *)(uintptr_t)to_i64(env, argv[0]), name);
  if (!p) { napi_throw_error(env, NULL, "bunffi: symbol not found"); return NULL; }
  napi_value out; CHECK(napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)p, &out));
  return out;
}

typedef int64_t (*tramp_t)(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, double, double, double, double);

/* call(fnptr, intArgs[], dblArgs[], ret) with ret in {"i32","i64","ptr","void"} */
static napi_value fn_call(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  int64_t a[8] = {0}; double d[4] = {0};
  uint32_t ni = 0, nd = 0;
  CHECK(napi_get_array_length(env, argv[1], &ni));
  CHECK(napi_get_array_length(env, argv[2], &nd));
  for (uint32_t i = 0; i < ni && i < 8; i++) { napi_value v; CHECK(napi_get_element(env, argv[1], i, &v)); a[i] = to_i64(env, v); }
  for (uint32_t i = 0; i < nd && i < 4; i++) { napi_value v; CHECK(napi_get_element(env, argv[2], i, &v)); CHECK(napi_get_value_double(env, v, &d[i])); }
  char ret[8]; size_t n = 0;
  CHECK(napi_get_value_string_utf8(env, argv[3], ret, sizeof ret, &n));
  tramp_t f = (tramp_t)(uintptr_t)to_i64(env, argv[0]);
  const int64_t r = f(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], d[0], d[1], d[2], d[3]);
  napi_value out;
  if (!strcmp(ret, "i32")) CHECK(napi_create_int32(env, (int32_t)r, &out));
  else if (!strcmp(ret, "void")) CHECK(napi_get_undefined(env, &out));
  else CHECK(napi_create_bigint_int64(env, r, &out));
  return out;
}

/* addressOf(ArrayBuffer | TypedArray | DataView) -> BigInt address of its first byte */
static napi_value fn_address_of(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void *data = NULL; size_t len = 0; bool is;
  CHECK(napi_is_arraybuffer(env, argv[0], &is));
  if (is) CHECK(napi_get_arraybuffer_info(env, argv[0], &data, &len));
  else {
    CHECK(napi_is_typedarray(env, argv[0], &is));
    if (is) { napi_typedarray_type ty; napi_value ab; size_t off; CHECK(napi_get_typedarray_info(env, argv[0], &ty, &len, &data, &ab, &off)); }
    else {
      CHECK(napi_is_dataview(env, argv[0], &is));
      if (!is) { napi_throw_type_error(env, NULL, "bunffi.ptr: expected an ArrayBuffer or a view"); return NULL; }
      napi_value ab; size_t off; CHECK(napi_get_dataview_info(env, argv[0], &len, &data, &ab, &off));
    }
  }
  napi_value out; CHECK(napi_create_bigint_uint64(env, (uint64_t)(uintptr_t)data, &out));
  return out;
}

static napi_value fn_cstring(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  CHECK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  const char *p = (const char *)(uintptr_t)to_i64(env, argv[0]);
  napi_value out; CHECK(napi_create_string_utf8(env, p ? p : "", NAPI_AUTO_LENGTH, &out));
  return out;
}

static napi_value init(napi_env env, napi_value exports) {
  napi_property_descriptor d[] = {
    {"dlopen", NULL, fn_dlopen, NULL, NULL, NULL, napi_default, NULL}, {"dlsym", NULL, fn_dlsym, NULL, NULL, NULL, napi_default, NULL},
    {"call", NULL, fn_call, NULL, NULL, NULL, napi_default, NULL}, {"addressOf", NULL, fn_address_of, NULL, NULL, NULL, napi_default, NULL},
    {"cstring", NULL, fn_cstring, NULL, NULL, NULL, napi_default, NULL}};
  napi_define_properties(env, exports, sizeof d / sizeof d[0], d);
  return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
