;;;; semz.decompress.gpu -- the MI355X batched decode engine behind semz.decompress.
;;;;
;;;; Load this system after (or instead of) "semz.decompress": it adds :around
;;;; methods to the reference's format protocol (common.lisp:262-289) and nothing
;;;; else.  No file of the reference is edited.  Needs libsemz_gpu.so (built by
;;;; `make -C decompress_amd/csrc`) on the foreign library search path, e.g.
;;;;   (pushnew #p"/path/to/repo/decompress_amd/" cffi:*foreign-library-directories*)
(asdf:defsystem "semz.decompress.gpu"
  :description "Batched GPU decoding for semz.decompress: whole gzip / bzip2 / xz inputs in one device call."
  :version "0.2.0"
  :licence "MIT"
  :depends-on ("semz.decompress" "cffi")
  :components ((:file "gpu")))
