/*
 * JNI glue between Gulon's Scala host code and libgulon_hip.so (include/gulon_hip.h).
 * NOT compiled in this repository's CI: the build image has no JDK (no jni.h).
 * On a box with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       gulon_jni.c -L../../gulon_amd/lib -lgulon_hip -o libgulon_jni.so
 * Java names: the natives are `@native def`s of the Scala `object net.tixxit.gulon.hip.Native`
 * (../scala/.../Native.scala).  scalac puts a native method of an object on its MODULE CLASS `Native$` -- the static
 * forwarders of `Native` call `Native$.MODULE$.f(...)` -- so the JVM resolves
 *   Java_net_tixxit_gulon_hip_Native_00024_<method>(JNIEnv *, jobject self, ...)
 * (`$` is mangled as `_00024`; an instance method: the receiver is the module instance, a jobject, not a jclass).
 * tests/test_integration_docs.py keeps the three lists -- this file, Native.scala and the table of INTEGRATION.md --
 * identical, and checks every native's parameter count and JNI types against its Scala signature.
 * Error mapping (include/gulon_hip.h): INVALID_ARGUMENT -> IllegalArgumentException (the reference's `require`),
 * ILLEGAL_STATE -> IllegalStateException, UNSUPPORTED -> UnsupportedOperationException, OOM -> OutOfMemoryError,
 * else RuntimeException.  A null array / buffer from the JVM side throws IllegalArgumentException before the
 * native call; a failed Get*ArrayElements throws OutOfMemoryError.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include "gulon_hip.h"

#define NAT(ret, name) JNIEXPORT ret JNICALL Java_net_tixxit_gulon_hip_Native_00024_##name

static void throw_named(JNIEnv *env, const char *cls, const char *msg) {
  jclass c = (*env)->FindClass(env, cls);
  if (c) (*env)->ThrowNew(env, c, msg);
}
static int throw_status(JNIEnv *env, int32_t rc) {
  if (rc == GULON_OK) return 0;
  const char *cls = rc == GULON_ERR_INVALID_ARGUMENT ? "java/lang/IllegalArgumentException"
                  : rc == GULON_ERR_ILLEGAL_STATE    ? "java/lang/IllegalStateException"
                  : rc == GULON_ERR_UNSUPPORTED      ? "java/lang/UnsupportedOperationException"
                  : rc == GULON_ERR_OOM              ? "java/lang/OutOfMemoryError"
                                                     : "java/lang/RuntimeException";
  throw_named(env, cls, gulon_last_error());
  return 1;
}

/* Pinned views of Java arrays: every array is fetched up front, checked, and released on every path. */
typedef struct { jarray arr; void *p; int kind; jint mode; } pin_t;   /* kind: 0 float, 1 int, 2 byte */
static int pin(JNIEnv *env, pin_t *t, jarray a, int kind, jint release_mode, int nullable) {
  t->arr = a; t->p = NULL; t->kind = kind; t->mode = release_mode;
  if (!a) {
    if (!nullable) throw_named(env, "java/lang/IllegalArgumentException", "null array");
    return nullable ? 0 : 1;
  }
  t->p = kind == 0 ? (void *)(*env)->GetFloatArrayElements(env, (jfloatArray)a, NULL)
       : kind == 1 ? (void *)(*env)->GetIntArrayElements(env, (jintArray)a, NULL)
                   : (void *)(*env)->GetByteArrayElements(env, (jbyteArray)a, NULL);
  if (!t->p) { throw_named(env, "java/lang/OutOfMemoryError", "Get*ArrayElements failed"); return 1; }
  return 0;
}
static void unpin(JNIEnv *env, pin_t *t) {
  if (!t->arr || !t->p) return;
  if (t->kind == 0) (*env)->ReleaseFloatArrayElements(env, (jfloatArray)t->arr, (jfloat *)t->p, t->mode);
  else if (t->kind == 1) (*env)->ReleaseIntArrayElements(env, (jintArray)t->arr, (jint *)t->p, t->mode);
  else (*env)->ReleaseByteArrayElements(env, (jbyteArray)t->arr, (jbyte *)t->p, t->mode);
  t->p = NULL;
}
#define IN JNI_ABORT   /* read-only input: no copy back */
#define OUT 0          /* output: copy back and free */
#define DS(h) ((gulon_dataset *)(intptr_t)(h))

/* ---- Matrix (Matrix.scala:3): Matrix.data flattened by the Scala side into one direct FloatBuffer ---- */
NAT(jlong, datasetCreate)(JNIEnv *env, jobject self, jobject buf, jint n, jint d) {
  gulon_dataset *ds = NULL;
  const float *x = buf ? (const float *)(*env)->GetDirectBufferAddress(env, buf) : NULL;
  if (!x) { throw_named(env, "java/lang/IllegalArgumentException", "expected a direct FloatBuffer"); return 0; }
  if (throw_status(env, gulon_dataset_create(x, n, d, &ds))) return 0;
  return (jlong)(intptr_t)ds;
}
NAT(void, datasetDestroy)(JNIEnv *env, jobject self, jlong h) { gulon_dataset_destroy(DS(h)); }

/* ---- KMeans (KMeans.scala) ---- */
/* KMeans.init (KMeans.scala:188-196) */
NAT(void, kmeansInit)(JNIEnv *env, jobject self, jlong ds, jint from, jint s, jint k, jint seed, jfloatArray centsOut) {
  pin_t co;
  if (pin(env, &co, centsOut, 0, OUT, 0)) return;
  int32_t rc = gulon_kmeans_init(DS(ds), from, s, k, seed, (float *)co.p, NULL);
  unpin(env, &co);
  throw_status(env, rc);
}
/* KMeans.assign / parAssign (KMeans.scala:18-22,57-98): rngBatch 0 = serial stream, 25000 = parAssign. */
NAT(void, kmeansAssign)(JNIEnv *env, jobject self, jlong ds, jint from, jint s, jfloatArray centroids, jint k, jint rngBatch,
                        jintArray assignments) {
  pin_t ce, as;
  if (pin(env, &ce, centroids, 0, IN, 0)) return;
  if (pin(env, &as, assignments, 1, OUT, 0)) { unpin(env, &ce); return; }
  int32_t rc = gulon_kmeans_assign(DS(ds), from, s, (const float *)ce.p, k, rngBatch, (int32_t *)as.p);
  unpin(env, &ce); unpin(env, &as);
  throw_status(env, rc);
}
/* KMeans.fromAssignment (KMeans.scala:198-226) */
NAT(void, kmeansUpdate)(JNIEnv *env, jobject self, jlong ds, jint from, jint s, jint k, jintArray assignments,
                        jfloatArray centsOut) {
  pin_t as, co;
  if (pin(env, &as, assignments, 1, IN, 0)) return;
  if (pin(env, &co, centsOut, 0, OUT, 0)) { unpin(env, &as); return; }
  int32_t rc = gulon_kmeans_update(DS(ds), from, s, k, (const int32_t *)as.p, (float *)co.p);
  unpin(env, &as); unpin(env, &co);
  throw_status(env, rc);
}
/* KMeans#iterate (KMeans.scala:100-106) */
NAT(void, kmeansIterate)(JNIEnv *env, jobject self, jlong ds, jint from, jint s, jfloatArray centsIn, jint k, jint iters,
                         jfloatArray centsOut) {
  pin_t ci, co;
  if (pin(env, &ci, centsIn, 0, IN, 0)) return;
  if (pin(env, &co, centsOut, 0, OUT, 0)) { unpin(env, &ci); return; }
  int32_t rc = gulon_kmeans_iterate(DS(ds), from, s, (const float *)ci.p, k, iters, (float *)co.p);
  unpin(env, &ci); unpin(env, &co);
  throw_status(env, rc);
}
/* ProgressReports (KMeans.scala:119-127) come back as two parallel arrays so that Scala can replay config.report
 * (KMeans.scala:150-151): ints {numIterations, converged, count} x reports, floats {mean, s} x reports. */
static void unpack_reports(const gulon_kmeans_report *reps, int n, jint *ri, jfloat *rf) {
  for (int i = 0; i < n; i++) {
    ri[3 * i] = reps[i].num_iterations; ri[3 * i + 1] = reps[i].converged; ri[3 * i + 2] = reps[i].step_count;
    rf[2 * i] = reps[i].step_mean; rf[2 * i + 1] = reps[i].step_s;
  }
}
/* KMeans.computeClusters (KMeans.scala:134-157) */
NAT(void, kmeansTrain)(JNIEnv *env, jobject self, jlong ds, jint from, jint s, jint k, jint maxIterations, jint seed,
                       jfloatArray centsOut, jintArray reportInts, jfloatArray reportFloats, jint maxReports,
                       jintArray nReports) {
  pin_t co, ri, rf, nr;
  if (maxReports < 0) { throw_named(env, "java/lang/IllegalArgumentException", "maxReports < 0"); return; }
  gulon_kmeans_report *reps = (gulon_kmeans_report *)malloc(sizeof(gulon_kmeans_report) * (size_t)(maxReports + 1));
  if (!reps) { throw_named(env, "java/lang/OutOfMemoryError", "malloc"); return; }
  if (pin(env, &co, centsOut, 0, OUT, 0)) { free(reps); return; }
  if (pin(env, &ri, reportInts, 1, OUT, 0)) { unpin(env, &co); free(reps); return; }
  if (pin(env, &rf, reportFloats, 0, OUT, 0)) { unpin(env, &co); unpin(env, &ri); free(reps); return; }
  if (pin(env, &nr, nReports, 1, OUT, 0)) { unpin(env, &co); unpin(env, &ri); unpin(env, &rf); free(reps); return; }
  int32_t n = 0;
  int32_t rc = gulon_kmeans_train(DS(ds), from, s, k, maxIterations, seed, (float *)co.p, reps, maxReports, &n);
  if (rc == GULON_OK) { ((jint *)nr.p)[0] = n; unpack_reports(reps, n < maxReports ? n : maxReports, (jint *)ri.p, (jfloat *)rf.p); }
  unpin(env, &co); unpin(env, &ri); unpin(env, &rf); unpin(env, &nr);
  free(reps);
  throw_status(env, rc);
}

/* ---- ProductQuantizer (ProductQuantizer.scala) ---- */
/* ProductQuantizer.apply (ProductQuantizer.scala:150-153): k*d floats out, quantizer j at k*from_j. */
NAT(void, pqTrain)(JNIEnv *env, jobject self, jlong ds, jint m, jint k, jint maxIterations, jfloatArray centsOut,
                   jintArray reportInts, jfloatArray reportFloats, jint maxReports, jintArray nReports) {
  pin_t co, ri, rf, nr;
  if (maxReports < 0 || m < 1) { throw_named(env, "java/lang/IllegalArgumentException", "bad report shape"); return; }
  const size_t total = (size_t)m * (size_t)maxReports;
  gulon_kmeans_report *reps = (gulon_kmeans_report *)malloc(sizeof(gulon_kmeans_report) * (total + 1));
  if (!reps) { throw_named(env, "java/lang/OutOfMemoryError", "malloc"); return; }
  if (pin(env, &co, centsOut, 0, OUT, 0)) { free(reps); return; }
  if (pin(env, &ri, reportInts, 1, OUT, 0)) { unpin(env, &co); free(reps); return; }
  if (pin(env, &rf, reportFloats, 0, OUT, 0)) { unpin(env, &co); unpin(env, &ri); free(reps); return; }
  if (pin(env, &nr, nReports, 1, OUT, 0)) { unpin(env, &co); unpin(env, &ri); unpin(env, &rf); free(reps); return; }
  int32_t rc = gulon_pq_train(DS(ds), m, k, maxIterations, (float *)co.p, reps, maxReports, (int32_t *)nr.p);
  if (rc == GULON_OK) unpack_reports(reps, (int)total, (jint *)ri.p, (jfloat *)rf.p);
  unpin(env, &co); unpin(env, &ri); unpin(env, &rf); unpin(env, &nr);
  free(reps);
  throw_status(env, rc);
}
/* ProductQuantizer.encode (ProductQuantizer.scala:25-35): m packed code arrays back to back. */
NAT(void, pqEncode)(JNIEnv *env, jobject self, jlong ds, jint m, jint k, jfloatArray cents, jbyteArray codesOut) {
  pin_t ce, co;
  if (pin(env, &ce, cents, 0, IN, 0)) return;
  if (pin(env, &co, codesOut, 2, OUT, 0)) { unpin(env, &ce); return; }
  int32_t rc = gulon_pq_encode(DS(ds), m, k, (const float *)ce.p, (uint8_t *)co.p);
  unpin(env, &ce); unpin(env, &co);
  throw_status(env, rc);
}

/* ---- Index (Index.scala) ---- */
/* Index.prepareQuery (Index.scala:352-383): tablesOut[B][m][k] */
NAT(void, prepareQuery)(JNIEnv *env, jobject self, jfloatArray cents, jint d, jint m, jint k, jfloatArray queries, jint b,
                        jfloatArray tablesOut) {
  pin_t ce, q, t;
  if (pin(env, &ce, cents, 0, IN, 0)) return;
  if (pin(env, &q, queries, 0, IN, 0)) { unpin(env, &ce); return; }
  if (pin(env, &t, tablesOut, 0, OUT, 0)) { unpin(env, &ce); unpin(env, &q); return; }
  int32_t rc = gulon_prepare_query((const float *)ce.p, d, m, k, (const float *)q.p, b, (float *)t.p);
  unpin(env, &ce); unpin(env, &q); unpin(env, &t);
  throw_status(env, rc);
}
/* PQIndex(productQuantizer, data) (Index.scala:385-391). */
NAT(jlong, indexCreate)(JNIEnv *env, jobject self, jbyteArray codes, jint n, jint d, jint m, jint k, jfloatArray cents,
                        jint rowBase) {
  gulon_index *ix = NULL;
  pin_t co, ce;
  if (pin(env, &co, codes, 2, IN, 0)) return 0;
  if (pin(env, &ce, cents, 0, IN, 0)) { unpin(env, &co); return 0; }
  int32_t rc = gulon_index_create((const uint8_t *)co.p, n, d, m, k, (const float *)ce.p, rowBase, &ix);
  unpin(env, &co); unpin(env, &ce);
  if (throw_status(env, rc)) return 0;
  return (jlong)(intptr_t)ix;
}
NAT(void, indexDestroy)(JNIEnv *env, jobject self, jlong h) { gulon_index_destroy((gulon_index *)(intptr_t)h); }
/* another workspace over the same device-resident codes (one per querying thread / batch in flight) */
NAT(jlong, indexContextCreate)(JNIEnv *env, jobject self, jlong h) {
  gulon_index *ctx = NULL;
  if (throw_status(env, gulon_index_context_create((gulon_index *)(intptr_t)h, &ctx))) return 0;
  return (jlong)(intptr_t)ctx;
}
static void query_common(JNIEnv *env, int which, jlong h, jfloatArray queries, jint b, jint k, jint a0, jint a1,
                         jintArray outIdx, jfloatArray outDist, jintArray outCount, jintArray outFlags) {
  pin_t q, oi, od, oc, of;
  if (pin(env, &q, queries, 0, IN, 0)) return;
  if (pin(env, &oi, outIdx, 1, OUT, 0)) { unpin(env, &q); return; }
  if (pin(env, &od, outDist, 0, OUT, 0)) { unpin(env, &q); unpin(env, &oi); return; }
  if (pin(env, &oc, outCount, 1, OUT, 0)) { unpin(env, &q); unpin(env, &oi); unpin(env, &od); return; }
  if (pin(env, &of, outFlags, 1, OUT, 1)) { unpin(env, &q); unpin(env, &oi); unpin(env, &od); unpin(env, &oc); return; }
  int32_t rc;
  if (which == 0)
    rc = gulon_index_batch_query((gulon_index *)(intptr_t)h, (const float *)q.p, b, k, a0, a1, (int32_t *)oi.p,
                                 (float *)od.p, (int32_t *)oc.p, (int32_t *)of.p);
  else if (which == 1)
    rc = gulon_sharded_index_batch_query((gulon_sharded_index *)(intptr_t)h, (const float *)q.p, b, k, (int32_t *)oi.p,
                                         (float *)od.p, (int32_t *)oc.p, (int32_t *)of.p);
  else if (which == 2)
    rc = gulon_exact_knn(DS(h), a0, a1, (const float *)q.p, b, k, (int32_t *)oi.p, (float *)od.p, (int32_t *)oc.p,
                         (int32_t *)of.p);
  else
    rc = gulon_grouped_index_batch_query((gulon_grouped_index *)(intptr_t)h, (const float *)q.p, b, k, a0, a1,
                                         (int32_t *)oi.p, (float *)od.p, (int32_t *)oc.p);
  unpin(env, &q); unpin(env, &oi); unpin(env, &od); unpin(env, &oc); unpin(env, &of);
  throw_status(env, rc);
}
/* PQIndex.batchQuery(k, vectors, from, until) (Index.scala:417-440) + Result.fromHeap (:83-94). */
NAT(void, indexBatchQuery)(JNIEnv *env, jobject self, jlong h, jfloatArray queries, jint b, jint k, jint from, jint until,
                           jintArray outIdx, jfloatArray outDist, jintArray outCount, jintArray outFlags) {
  query_common(env, 0, h, queries, b, k, from, until, outIdx, outDist, outCount, outFlags);
}
/* Index.exactNearestNeighbours (Index.scala:209-229) for B queries over rows [from, until) */
NAT(void, exactKnn)(JNIEnv *env, jobject self, jlong ds, jint from, jint until, jfloatArray queries, jint b, jint k,
                    jintArray outIdx, jfloatArray outDist, jintArray outCount, jintArray outFlags) {
  query_common(env, 2, ds, queries, b, k, from, until, outIdx, outDist, outCount, outFlags);
}
/* TopKHeap.merge across partial lists (TopKHeap.scala:44-53): lists [lists][B][K+1], ascending, (+inf, MAX) padded */
NAT(void, topkMerge)(JNIEnv *env, jobject self, jfloatArray partDist, jintArray partIdx, jint lists, jint b, jint k,
                     jintArray outIdx, jfloatArray outDist, jintArray outCount, jintArray outFlags) {
  pin_t pd, pi, oi, od, oc, of;
  if (pin(env, &pd, partDist, 0, IN, 0)) return;
  if (pin(env, &pi, partIdx, 1, IN, 0)) { unpin(env, &pd); return; }
  if (pin(env, &oi, outIdx, 1, OUT, 0)) { unpin(env, &pd); unpin(env, &pi); return; }
  if (pin(env, &od, outDist, 0, OUT, 0)) { unpin(env, &pd); unpin(env, &pi); unpin(env, &oi); return; }
  if (pin(env, &oc, outCount, 1, OUT, 1)) { unpin(env, &pd); unpin(env, &pi); unpin(env, &oi); unpin(env, &od); return; }
  if (pin(env, &of, outFlags, 1, OUT, 1)) { unpin(env, &pd); unpin(env, &pi); unpin(env, &oi); unpin(env, &od); unpin(env, &oc); return; }
  int32_t rc = gulon_topk_merge((const float *)pd.p, (const int32_t *)pi.p, lists, b, k, (int32_t *)oi.p, (float *)od.p,
                                (int32_t *)oc.p, (int32_t *)of.p);
  unpin(env, &pd); unpin(env, &pi); unpin(env, &oi); unpin(env, &od); unpin(env, &oc); unpin(env, &of);
  throw_status(env, rc);
}

/* ---- PQIndex row-sharded over the GPUs of the node, inside this one JVM (sharded.hip: RCCL all-gathers) ---- */
NAT(jlong, shardedIndexCreate)(JNIEnv *env, jobject self, jbyteArray codes, jint n, jint d, jint m, jint k, jfloatArray cents,
                               jintArray devices) {
  gulon_sharded_index *sx = NULL;
  pin_t co, ce, dv;
  if (pin(env, &co, codes, 2, IN, 0)) return 0;
  if (pin(env, &ce, cents, 0, IN, 0)) { unpin(env, &co); return 0; }
  if (pin(env, &dv, devices, 1, IN, 0)) { unpin(env, &co); unpin(env, &ce); return 0; }
  const jsize shards = (*env)->GetArrayLength(env, devices);
  int32_t rc = gulon_sharded_index_create((const uint8_t *)co.p, n, d, m, k, (const float *)ce.p, (const int32_t *)dv.p,
                                          (int32_t)shards, &sx);
  unpin(env, &co); unpin(env, &ce); unpin(env, &dv);
  if (throw_status(env, rc)) return 0;
  return (jlong)(intptr_t)sx;
}
NAT(void, shardedIndexDestroy)(JNIEnv *env, jobject self, jlong h) { gulon_sharded_index_destroy((gulon_sharded_index *)(intptr_t)h); }
NAT(void, shardedIndexBatchQuery)(JNIEnv *env, jobject self, jlong h, jfloatArray queries, jint b, jint k, jintArray outIdx,
                                  jfloatArray outDist, jintArray outCount, jintArray outFlags) {
  query_common(env, 1, h, queries, b, k, 0, 0, outIdx, outDist, outCount, outFlags);
}

/* ---- GroupedIndex (Index.scala:231-308) ---- */
NAT(jlong, groupResiduals)(JNIEnv *env, jobject self, jlong ds, jintArray perm, jintArray groupOf, jfloatArray centroids, jint g) {
  gulon_dataset *out = NULL;
  pin_t p, go, ce;
  if (pin(env, &p, perm, 1, IN, 0)) return 0;
  if (pin(env, &go, groupOf, 1, IN, 0)) { unpin(env, &p); return 0; }
  if (pin(env, &ce, centroids, 0, IN, 0)) { unpin(env, &p); unpin(env, &go); return 0; }
  int32_t rc = gulon_dataset_group_residuals(DS(ds), (const int32_t *)p.p, (const int32_t *)go.p, (const float *)ce.p, g, &out);
  unpin(env, &p); unpin(env, &go); unpin(env, &ce);
  if (throw_status(env, rc)) return 0;
  return (jlong)(intptr_t)out;
}
NAT(jlong, groupedIndexCreate)(JNIEnv *env, jobject self, jbyteArray codes, jint n, jint d, jint m, jint k, jfloatArray pqCents,
                               jfloatArray groupCents, jintArray offsets, jint g) {
  gulon_grouped_index *out = NULL;
  pin_t co, pc, gc, of;
  if (pin(env, &co, codes, 2, IN, 0)) return 0;
  if (pin(env, &pc, pqCents, 0, IN, 0)) { unpin(env, &co); return 0; }
  if (pin(env, &gc, groupCents, 0, IN, 0)) { unpin(env, &co); unpin(env, &pc); return 0; }
  if (pin(env, &of, offsets, 1, IN, 0)) { unpin(env, &co); unpin(env, &pc); unpin(env, &gc); return 0; }
  int32_t rc = gulon_grouped_index_create((const uint8_t *)co.p, n, d, m, k, (const float *)pc.p, (const float *)gc.p,
                                          (const int32_t *)of.p, g, &out);
  unpin(env, &co); unpin(env, &pc); unpin(env, &gc); unpin(env, &of);
  if (throw_status(env, rc)) return 0;
  return (jlong)(intptr_t)out;
}
NAT(void, groupedIndexDestroy)(JNIEnv *env, jobject self, jlong h) { gulon_grouped_index_destroy((gulon_grouped_index *)(intptr_t)h); }
/* strategy 0 = LimitGroups, 1 = LimitVectors */
NAT(void, groupedIndexBatchQuery)(JNIEnv *env, jobject self, jlong h, jfloatArray queries, jint b, jint k, jint strategy,
                                  jint limit, jintArray outIdx, jfloatArray outDist, jintArray outCount) {
  query_common(env, 3, h, queries, b, k, strategy, limit, outIdx, outDist, outCount, NULL);
}
