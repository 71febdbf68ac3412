/*
 * tests/stubs/jni.h -- a TEST DOUBLE of the JDK's <jni.h>, just wide enough to compile the
 * reference's UNCHANGED JNI glue (src/smatrix_jni.c) in an image without a JDK, so that
 * tests/test_binding_link.py can check that the glue links against this repo's smatrix.o and
 * resolves exactly the eight public smatrix_* symbols.
 * Written from the JNI specification's type and function-table names; it is not used by the
 * product and never by the oracle.
 */
#ifndef SMX_TEST_JNI_H
#define SMX_TEST_JNI_H
#include <stdarg.h>
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef uint8_t jboolean;
typedef void* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef void* jfieldID;
typedef void* jmethodID;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv*, const char*);
  jint (*ThrowNew)(JNIEnv*, jclass, const char*);
  jclass (*GetObjectClass)(JNIEnv*, jobject);
  jmethodID (*GetMethodID)(JNIEnv*, jclass, const char*, const char*);
  void (*CallVoidMethod)(JNIEnv*, jobject, jmethodID, ...);
  jfieldID (*GetFieldID)(JNIEnv*, jclass, const char*, const char*);
  jlong (*GetLongField)(JNIEnv*, jobject, jfieldID);
  void (*SetLongField)(JNIEnv*, jobject, jfieldID, jlong);
  const char* (*GetStringUTFChars)(JNIEnv*, jstring, jboolean*);
  void (*ReleaseStringUTFChars)(JNIEnv*, jstring, const char*);
};
#endif
