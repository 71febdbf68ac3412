/*
 * tests/stubs/ruby/ruby.h -- a TEST DOUBLE of MRI's <ruby/ruby.h>, just wide enough to compile the
 * reference's UNCHANGED Ruby glue (src/smatrix_ruby.c, src/smatrix_ruby.h) in an image without Ruby,
 * so that tests/test_binding_link.py can check that the glue links against this repo's smatrix.o the
 * way src/ruby/Makefile:18-19 links it (glue + ../smatrix.o, interpreter symbols left to the loader)
 * and needs nothing but the public smatrix_* symbols.
 * Written from the C-API names the glue uses (smatrix_ruby.c:15-174); not used by the product or the oracle.
 */
#ifndef SMX_TEST_RUBY_H
#define SMX_TEST_RUBY_H
#include <stdint.h>

typedef uintptr_t VALUE;
typedef uintptr_t ID;

enum ruby_value_type {
  RUBY_T_NONE = 0x00, RUBY_T_STRING = 0x05, RUBY_T_DATA = 0x0c, RUBY_T_NIL = 0x11, RUBY_T_FIXNUM = 0x15
};
#define T_STRING RUBY_T_STRING

#define Qnil ((VALUE)8)

extern VALUE rb_cObject;
extern VALUE rb_eTypeError;

int rb_type(VALUE obj);
VALUE rb_iv_get(VALUE obj, const char* name);
VALUE rb_iv_set(VALUE obj, const char* name, VALUE val);
void rb_raise(VALUE exc, const char* fmt, ...);
VALUE rb_define_class(const char* name, VALUE super);
void rb_define_method(VALUE klass, const char* name, VALUE (*func)(), int argc);

char* rb_test_string_ptr(VALUE str);
#define RSTRING_PTR(s) rb_test_string_ptr(s)

VALUE rb_int2inum(intptr_t v);
long rb_num2int(VALUE v);
#define INT2NUM(v) rb_int2inum((intptr_t)(v))
#define NUM2INT(v) ((int)rb_num2int(v))

VALUE rb_data_object_wrap(VALUE klass, void* datap, void (*mark)(void*), void (*free_fn)(void*));
void* rb_data_object_get(VALUE obj);
#define Data_Wrap_Struct(klass, mark, free_fn, sval) \
  rb_data_object_wrap((klass), (sval), (void (*)(void*))(mark), (void (*)(void*))(free_fn))
#define Data_Get_Struct(obj, type, sval) ((sval) = (type*)rb_data_object_get(obj))

#endif
