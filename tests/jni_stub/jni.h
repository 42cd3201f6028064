/*
 * tests/jni_stub/jni.h -- NOT the JDK's jni.h.  A minimal stand-in that declares only the JNIEnv entries
 * integration/jni/imm3_jni.c uses, so that the shim can be SYNTAX-checked (gcc -fsyntax-only) in an image without a
 * JDK.  Nothing is linked or run against it; signatures follow the JNI specification.  Test infrastructure only.
 */
#ifndef IMM3_TEST_JNI_STUB_H
#define IMM3_TEST_JNI_STUB_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef double jdouble;
typedef jint jsize;
typedef void *jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
typedef jarray jbyteArray;
typedef jarray jobjectArray;
typedef uint8_t jboolean;

#define JNIEXPORT
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    void (*DeleteLocalRef)(JNIEnv *, jobject);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jobject (*GetObjectArrayElement)(JNIEnv *, jobjectArray, jsize);
    jint *(*GetIntArrayElements)(JNIEnv *, jintArray, jboolean *);
    jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *, jdoubleArray, jboolean *);
    void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
    void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
    void (*ReleaseDoubleArrayElements)(JNIEnv *, jdoubleArray, jdouble *, jint);
    void (*GetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, jbyte *);
    jlongArray (*NewLongArray)(JNIEnv *, jsize);
    void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
    void *(*GetDirectBufferAddress)(JNIEnv *, jobject);
    jlong (*GetDirectBufferCapacity)(JNIEnv *, jobject);
};
#endif
