/* tests/stubs/ruby/config.h -- TEST DOUBLE of <ruby/config.h> (empty on purpose): lets the reference's
 * UNCHANGED Ruby glue (src/smatrix_ruby.c:10) compile in an image without Ruby.  See ruby.h next to it. */
#ifndef SMX_TEST_RUBY_CONFIG_H
#define SMX_TEST_RUBY_CONFIG_H
#endif
