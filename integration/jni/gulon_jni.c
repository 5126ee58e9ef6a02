/*
 * JNI glue between Gulon's Scala host code and libgulon_hip.so (include/gulon_hip.h).
 * NOT compiled in this repository's CI: the build image has no JDK (no jni.h).
 * On a box with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       gulon_jni.c -L../../gulon_amd/lib -lgulon_hip -o libgulon_jni.so
 * Java names: net.tixxit.gulon.hip.Native (see ../scala/.../Native.scala).
 * Error mapping (include/gulon_hip.h): INVALID_ARGUMENT -> IllegalArgumentException
 * (the reference's `require`), ILLEGAL_STATE -> IllegalStateException, else RuntimeException.
 */
#include <jni.h>
#include <stdint.h>
#include "gulon_hip.h"

static int throw_status(JNIEnv *env, int32_t rc) {
  if (rc == GULON_OK) return 0;
  const char *cls = rc == GULON_ERR_INVALID_ARGUMENT ? "java/lang/IllegalArgumentException"
                  : rc == GULON_ERR_ILLEGAL_STATE    ? "java/lang/IllegalStateException"
                  : rc == GULON_ERR_UNSUPPORTED      ? "java/lang/UnsupportedOperationException"
                                                     : "java/lang/RuntimeException";
  (*env)->ThrowNew(env, (*env)->FindClass(env, cls), gulon_last_error());
  return 1;
}

/* Matrix.data flattened by the Scala side into one direct FloatBuffer (row-major, ld = cols). */
JNIEXPORT jlong JNICALL Java_net_tixxit_gulon_hip_Native_datasetCreate(JNIEnv *env, jclass c, jobject buf, jint n, jint d) {
  gulon_dataset *ds = NULL;
  const float *x = (const float *)(*env)->GetDirectBufferAddress(env, buf);
  if (throw_status(env, gulon_dataset_create(x, n, d, &ds))) return 0;
  return (jlong)(intptr_t)ds;
}
JNIEXPORT void JNICALL Java_net_tixxit_gulon_hip_Native_datasetDestroy(JNIEnv *env, jclass c, jlong h) {
  gulon_dataset_destroy((gulon_dataset *)(intptr_t)h);
}

/* ProductQuantizer.apply (ProductQuantizer.scala:150-153): returns k*d floats, quantizer j at k*from_j. */
JNIEXPORT void JNICALL Java_net_tixxit_gulon_hip_Native_pqTrain(JNIEnv *env, jclass c, jlong ds, jint m, jint k,
                                                               jint maxIterations, jfloatArray centsOut,
                                                               jintArray reportInts, jfloatArray reportFloats,
                                                               jint maxReports, jintArray nReports) {
  jfloat *cents = (*env)->GetFloatArrayElements(env, centsOut, NULL);
  jint *nrep = (*env)->GetIntArrayElements(env, nReports, NULL);
  /* reports come back as two parallel arrays so that Scala can replay config.report
   * (KMeans.scala:150-151): ints {numIterations, converged, count} x m x maxReports, floats {mean, s}. */
  gulon_kmeans_report *reps = (gulon_kmeans_report *)malloc(sizeof(gulon_kmeans_report) * (size_t)m * maxReports);
  int32_t rc = gulon_pq_train((gulon_dataset *)(intptr_t)ds, m, k, maxIterations, cents, reps, maxReports, (int32_t *)nrep);
  if (rc == GULON_OK) {
    jint *ri = (*env)->GetIntArrayElements(env, reportInts, NULL);
    jfloat *rf = (*env)->GetFloatArrayElements(env, reportFloats, NULL);
    for (int i = 0; i < m * maxReports; i++) {
      ri[3 * i] = reps[i].num_iterations; ri[3 * i + 1] = reps[i].converged; ri[3 * i + 2] = reps[i].step_count;
      rf[2 * i] = reps[i].step_mean; rf[2 * i + 1] = reps[i].step_s;
    }
    (*env)->ReleaseIntArrayElements(env, reportInts, ri, 0);
    (*env)->ReleaseFloatArrayElements(env, reportFloats, rf, 0);
  }
  free(reps);
  (*env)->ReleaseFloatArrayElements(env, centsOut, cents, 0);
  (*env)->ReleaseIntArrayElements(env, nReports, nrep, 0);
  throw_status(env, rc);
}

/* ProductQuantizer.encode (ProductQuantizer.scala:25-35): m packed code arrays back to back. */
JNIEXPORT void JNICALL Java_net_tixxit_gulon_hip_Native_pqEncode(JNIEnv *env, jclass c, jlong ds, jint m, jint k,
                                                                jfloatArray cents, jbyteArray codesOut) {
  jfloat *ce = (*env)->GetFloatArrayElements(env, cents, NULL);
  jbyte *co = (*env)->GetByteArrayElements(env, codesOut, NULL);
  int32_t rc = gulon_pq_encode((gulon_dataset *)(intptr_t)ds, m, k, ce, (uint8_t *)co);
  (*env)->ReleaseByteArrayElements(env, codesOut, co, 0);
  (*env)->ReleaseFloatArrayElements(env, cents, ce, JNI_ABORT);
  throw_status(env, rc);
}

/* PQIndex(productQuantizer, data) (Index.scala:385-391). */
JNIEXPORT jlong JNICALL Java_net_tixxit_gulon_hip_Native_indexCreate(JNIEnv *env, jclass c, jbyteArray codes, jint n,
                                                                    jint d, jint m, jint k, jfloatArray cents,
                                                                    jint rowBase) {
  gulon_index *ix = NULL;
  jbyte *co = (*env)->GetByteArrayElements(env, codes, NULL);
  jfloat *ce = (*env)->GetFloatArrayElements(env, cents, NULL);
  int32_t rc = gulon_index_create((const uint8_t *)co, n, d, m, k, ce, rowBase, &ix);
  (*env)->ReleaseByteArrayElements(env, codes, co, JNI_ABORT);
  (*env)->ReleaseFloatArrayElements(env, cents, ce, JNI_ABORT);
  if (throw_status(env, rc)) return 0;
  return (jlong)(intptr_t)ix;
}
JNIEXPORT void JNICALL Java_net_tixxit_gulon_hip_Native_indexDestroy(JNIEnv *env, jclass c, jlong h) {
  gulon_index_destroy((gulon_index *)(intptr_t)h);
}

/* PQIndex.batchQuery(k, vectors, from, until) (Index.scala:417-440) + Result.fromHeap (:83-94). */
JNIEXPORT void JNICALL Java_net_tixxit_gulon_hip_Native_indexBatchQuery(JNIEnv *env, jclass c, jlong h, jfloatArray queries,
                                                                       jint b, jint k, jint from, jint until,
                                                                       jintArray outIdx, jfloatArray outDist,
                                                                       jintArray outCount, jintArray outFlags) {
  jfloat *q = (*env)->GetFloatArrayElements(env, queries, NULL);
  jint *oi = (*env)->GetIntArrayElements(env, outIdx, NULL);
  jfloat *od = (*env)->GetFloatArrayElements(env, outDist, NULL);
  jint *oc = (*env)->GetIntArrayElements(env, outCount, NULL);
  jint *of = (*env)->GetIntArrayElements(env, outFlags, NULL);
  int32_t rc = gulon_index_batch_query((gulon_index *)(intptr_t)h, q, b, k, from, until, (int32_t *)oi, od,
                                       (int32_t *)oc, (int32_t *)of);
  (*env)->ReleaseFloatArrayElements(env, queries, q, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, outIdx, oi, 0);
  (*env)->ReleaseFloatArrayElements(env, outDist, od, 0);
  (*env)->ReleaseIntArrayElements(env, outCount, oc, 0);
  (*env)->ReleaseIntArrayElements(env, outFlags, of, 0);
  throw_status(env, rc);
}

/* KMeans.assign / parAssign (KMeans.scala:18-22,57-98): rngBatch 0 = serial stream, 25000 = parAssign. */
JNIEXPORT void JNICALL Java_net_tixxit_gulon_hip_Native_kmeansAssign(JNIEnv *env, jclass c, jlong ds, jint from, jint s,
                                                                    jfloatArray centroids, jint k, jint rngBatch,
                                                                    jintArray assignments) {
  jfloat *ce = (*env)->GetFloatArrayElements(env, centroids, NULL);
  jint *as = (*env)->GetIntArrayElements(env, assignments, NULL);
  int32_t rc = gulon_kmeans_assign((gulon_dataset *)(intptr_t)ds, from, s, ce, k, rngBatch, (int32_t *)as);
  (*env)->ReleaseFloatArrayElements(env, centroids, ce, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, assignments, as, 0);
  throw_status(env, rc);
}
/* kmeansInit / kmeansUpdate / kmeansIterate / kmeansTrain / exactKnn / prepareQuery follow the same pattern. */

/* GroupedIndex (Index.scala:231-308) */
JNIEXPORT jlong JNICALL Java_net_tixxit_gulon_hip_Native_groupResiduals(JNIEnv *env, jclass c, jlong ds, jintArray perm,
                                                                      jintArray groupOf, jfloatArray centroids, jint g) {
  gulon_dataset *out = NULL;
  jint *p = (*env)->GetIntArrayElements(env, perm, NULL), *go = (*env)->GetIntArrayElements(env, groupOf, NULL);
  jfloat *ce = (*env)->GetFloatArrayElements(env, centroids, NULL);
  int32_t rc = gulon_dataset_group_residuals((gulon_dataset *)(intptr_t)ds, (const int32_t *)p, (const int32_t *)go, ce, g, &out);
  (*env)->ReleaseIntArrayElements(env, perm, p, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, groupOf, go, JNI_ABORT);
  (*env)->ReleaseFloatArrayElements(env, centroids, ce, JNI_ABORT);
  if (throw_status(env, rc)) return 0;
  return (jlong)(intptr_t)out;
}
JNIEXPORT jlong JNICALL Java_net_tixxit_gulon_hip_Native_groupedIndexCreate(JNIEnv *env, jclass c, jbyteArray codes, jint n,
                                                                          jint d, jint m, jint k, jfloatArray pqCents,
                                                                          jfloatArray groupCents, jintArray offsets, jint g) {
  gulon_grouped_index *out = NULL;
  jbyte *co = (*env)->GetByteArrayElements(env, codes, NULL);
  jfloat *pc = (*env)->GetFloatArrayElements(env, pqCents, NULL), *gc = (*env)->GetFloatArrayElements(env, groupCents, NULL);
  jint *of = (*env)->GetIntArrayElements(env, offsets, NULL);
  int32_t rc = gulon_grouped_index_create((const uint8_t *)co, n, d, m, k, pc, gc, (const int32_t *)of, g, &out);
  (*env)->ReleaseByteArrayElements(env, codes, co, JNI_ABORT);
  (*env)->ReleaseFloatArrayElements(env, pqCents, pc, JNI_ABORT);
  (*env)->ReleaseFloatArrayElements(env, groupCents, gc, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, offsets, of, JNI_ABORT);
  if (throw_status(env, rc)) return 0;
  return (jlong)(intptr_t)out;
}
JNIEXPORT void JNICALL Java_net_tixxit_gulon_hip_Native_groupedIndexDestroy(JNIEnv *env, jclass c, jlong h) {
  gulon_grouped_index_destroy((gulon_grouped_index *)(intptr_t)h);
}
JNIEXPORT void JNICALL Java_net_tixxit_gulon_hip_Native_groupedIndexBatchQuery(JNIEnv *env, jclass c, jlong h,
                                                                             jfloatArray queries, jint b, jint k,
                                                                             jint strategy, jint limit, jintArray outIdx,
                                                                             jfloatArray outDist, jintArray outCount) {
  jfloat *q = (*env)->GetFloatArrayElements(env, queries, NULL), *od = (*env)->GetFloatArrayElements(env, outDist, NULL);
  jint *oi = (*env)->GetIntArrayElements(env, outIdx, NULL), *oc = (*env)->GetIntArrayElements(env, outCount, NULL);
  int32_t rc = gulon_grouped_index_batch_query((gulon_grouped_index *)(intptr_t)h, q, b, k, strategy, limit, (int32_t *)oi,
                                               od, (int32_t *)oc);
  (*env)->ReleaseFloatArrayElements(env, queries, q, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, outIdx, oi, 0);
  (*env)->ReleaseFloatArrayElements(env, outDist, od, 0);
  (*env)->ReleaseIntArrayElements(env, outCount, oc, 0);
  throw_status(env, rc);
}
