;;;; gpu.lisp -- CFFI glue between semz.decompress and libsemz_gpu.so (include/semz_gpu.h).
;;;;
;;;; What it hooks.  The reference's "plugin API" for a format is three generic
;;;; functions (common.lisp:262-289):
;;;;
;;;;   byte-source->decompression-state (format byte-source &key)  -> state, header
;;;;   next-decompressed-chunk (state)                             -> chunk start end finalp
;;;;   make-reset-state (old-state)                                -> new-state, header | nil
;;;;
;;;; `make-decompression-stream' (interface.lisp:46-143, a plain function that is
;;;; left alone) calls the first one with the byte source it made from the
;;;; caller's input.  This file adds an :around method to it for every format
;;;; keyword.  When the byte source is the zero-copy wrapper of a `buffer'-typed
;;;; vector (array->buffer-stream, io.lisp:28-31) the method hands the whole
;;;; input to the engine, which decodes every member / block in one batch on the
;;;; GPU and keeps the result in device memory, and returns a `gpu-batch-state'.
;;;; Otherwise (streams, chunked sources, nested calls of the reference's own
;;;; methods, format options the engine does not take) it calls the next method:
;;;; the reference decodes as before.
;;;;
;;;; How the state behaves.  `next-decompressed-chunk' hands out the decoded
;;;; stream one member at a time, in chunks of bounded size fetched from the
;;;; device into one reusable Lisp vector, and says `finalp' at the end of each
;;;; member; `make-reset-state' moves on to the next member.  So the member loop
;;;; of interface.lisp:91-133 runs unchanged: a member's data is delivered before
;;;; anything that follows it is looked at, only the first member's header is
;;;; reported, single-member formats still get "Trailing data detected." from
;;;; the reference itself, and an error in member k is signalled by the read
;;;; that reaches member k, after members 0..k-1 have been delivered
;;;; (interface.lisp:105-108).
;;;;
;;;; Not executed in the build image of this repository (it holds no Common Lisp
;;;; implementation); tests/test_lisp_glue.py checks every symbol of the
;;;; reference and of the C ABI that this file uses against their sources, and
;;;; decompress_amd/api.py is the executed mirror of the same state machine.
(cl:defpackage #:semz.decompress.gpu
  (:use #:cl)
  (:export #:*enabled*
           #:*device*
           #:*minimum-input-size*
           #:*chunk-size*
           #:engine-error
           #:open-engine
           #:close-engine))
(cl:in-package #:semz.decompress.gpu)

;;;; Foreign interface (include/semz_gpu.h)

(cffi:define-foreign-library libsemz-gpu
  (:unix (:or "libsemz_gpu.so" "decompress_amd/libsemz_gpu.so"))
  (t (:default "libsemz_gpu")))

(cffi:defcstruct semz-result
  (status :int32)
  (detail :uint32)
  (in-end-bit :uint64)
  (out-len :uint64)
  (out-off :uint64)
  (check :uint8 :count 32)
  (check-len :uint32)
  (aux :uint32))

(cffi:defcstruct semz-stream-info
  (status :int32)
  (detail :uint32)
  (total-out :uint64)
  (first-header :uint32 :count 8)
  (nmembers :uint32)
  (reserved :uint32))

(cffi:defcstruct semz-member
  (in-end :uint64)
  (out-end :uint64))

(cffi:defcfun ("semz_open" semz-open) :int (device :int) (ctx :pointer))
(cffi:defcfun ("semz_open_multi" semz-open-multi) :int (devices :pointer) (ndev :int) (ctx :pointer))
(cffi:defcfun ("semz_close" semz-close) :void (ctx :pointer))
(cffi:defcfun ("semz_last_error" semz-last-error) :string (ctx :pointer))
(cffi:defcfun ("semz_strerror" semz-strerror) :string (status :int))
(cffi:defcfun ("semz_set_option" semz-set-option) :int (ctx :pointer) (name :string) (value :long))
(cffi:defcfun ("semz_get_option" semz-get-option) :int (ctx :pointer) (name :string) (value :pointer))
(cffi:defcfun ("semz_stream_decode" semz-stream-decode) :int
  (ctx :pointer) (fmt :int) (in :pointer) (in-len :size) (all-members :int)
  (decoded :pointer) (res :pointer) (info :pointer))
(cffi:defcfun ("semz_stream_members" semz-stream-members) :int
  (decoded :pointer) (members :pointer) (nmembers :pointer))
(cffi:defcfun ("semz_stream_read" semz-stream-read) :int
  (decoded :pointer) (offset :uint64) (dst :pointer) (len :size))
(cffi:defcfun ("semz_stream_release" semz-stream-release) :void (decoded :pointer))

;;; SEMZ_FMT_* of semz_gpu.h, in the order of the reference's format list (example.lisp:77-78)
(defparameter *format-codes*
  '(:deflate 0 :zlib 1 :gzip 2 :bzip2 3 :raw-lzma 4 :lzma 5 :raw-lzma2 6 :lzma2 7 :xz 8))

;;; formats whose `make-reset-state' starts another member (gzip.lisp:92-93, bzip2.lisp:453-454, xz.lisp:118-134)
(defparameter *multi-member-formats* '(:gzip :bzip2 :xz))

;;;; Engine

(defvar *enabled* t
  "When nil the :around methods step aside and the reference decodes everything.")
(defvar *device* 0
  "HIP device ordinal the engine opens, or a list of ordinals: then one input is decoded on all of them
(semz_open_multi; the first is the primary device).")
(defvar *minimum-input-size* 65536
  "Inputs shorter than this many octets are decoded by the reference: a device call does not pay for them.")
(defvar *chunk-size* (* 4 1024 1024)
  "Upper bound of the chunks `next-decompressed-chunk' hands out (common.lisp:255-257).")

(define-condition engine-error (error)
  ((message :initarg :message :reader engine-error-message))
  (:report (lambda (c s) (format s "semz.decompress.gpu: ~a" (engine-error-message c))))
  (:documentation "The engine itself failed (no GPU, HIP failure, API misuse). Never signalled for bad data:
that is `semz.decompress:decompression-error', as in the reference."))

(defvar *batch-budget* 0
  "Device bytes a predicted batch may take (the engine's option \"batch-budget\"); a larger one is decoded window
by window into host memory.  0 = three quarters of what the device has free when a call starts.")
(defvar *engine* nil "The semz_ctx pointer, opened on first use.")

(defun open-engine ()
  "Opens the engine on `*device*'.  There is no CPU decode path in the library: without a usable GPU this
signals `engine-error' (bind `*enabled*' to nil to use the reference's decoders)."
  (or *engine*
      (progn
        (cffi:use-foreign-library libsemz-gpu)
        (cffi:with-foreign-object (p :pointer)
          (let ((rc (if (listp *device*)
                        (cffi:with-foreign-object (devs :int (length *device*))
                          (loop for d in *device* for i from 0 do (setf (cffi:mem-aref devs :int i) d))
                          (semz-open-multi devs (length *device*) p))
                        (semz-open *device* p))))
            (unless (zerop rc)
              (error 'engine-error
                     :message (format nil "semz_open(~a) failed with ~d (no usable MI355X?)" *device* rc))))
          (setf *engine* (cffi:mem-ref p :pointer))
          (unless (zerop *batch-budget*)
            (semz-set-option *engine* "batch-budget" *batch-budget*))
          *engine*))))

(defun close-engine ()
  (when *engine*
    (semz-close *engine*)
    (setf *engine* nil)))

(defun check-rc (rc what)
  (unless (zerop rc)
    (error 'engine-error :message (format nil "~a failed with ~d: ~a" what rc (semz-last-error *engine*)))))

;;;; Conditions: one status per `die' site of the reference (include/semz_status.h)

;;; status -> the reference's own format-control string and where its arguments come from:
;;;   :detail        semz_result.detail, the first argument of the message
;;;   :aux           semz_result.aux of a failed stream: a second argument that is not a checksum
;;;   :check         semz_result.check, the checksum the decoder computed ("got"), as the integer the reference compares
;;;   :high-octet / :low-octet   the two octets packed into detail (bzip2.lisp:430)
;;;   :stored-check  the check field of the xz block as stored in the input; :hex-digits = twice its size (xz.lisp:452-454)
;;; Statuses that are not listed have no message in the reference (SEMZ_E_BZ_ORIGPTR, SEMZ_E_LZMA2_NO_STATE: see
;;; DESIGN.md) and are reported with the engine's text.
(defparameter *messages*
  '((10 "Underfull Huffman tree.")                                                        ; huffman.lisp:183
    (11 "Overfull Huffman tree.")                                                         ; huffman.lisp:184
    (20 "Block uses reserved BTYPE.")                                                     ; deflate.lisp:397
    (21 "Checksum mismatch in uncompressed block (required ~4,'0x, got ~4,'0x)." :aux :detail) ; deflate.lisp:384
    (22 "Code lengths expand beyond bounds.")                                             ; deflate.lisp:187
    (23 "Tried to repeat non-existent last code length.")                                 ; deflate.lisp:190
    (24 "Distance code out of bounds (0-29): ~d" :detail)                                 ; deflate.lisp:298
    (25 "Unique distance code is out of bounds (0-29) or is not ~
                        encoded as a zero bit.")                                          ; deflate.lisp:304-305
    (26 "Length code in literal-only block: ~d" :detail)                                  ; deflate.lisp:324
    (27 "Reference points back further (~d) than the window allows (~d)." :detail :aux)   ; deflate.lisp:330
    (28 "Invalid literal/length code: ~d" :detail)                                        ; deflate.lisp:336
    (40 "Unsupported compression method: ~X" :detail)                                     ; zlib.lisp:31
    (41 "Invalid compression info (must be 0 - 7): ~X" :detail)                           ; zlib.lisp:33
    (42 "Invalid flag checksum.")                                                         ; zlib.lisp:35
    (43 "Cannot use preset dictionaries without a dictionary function.")                  ; zlib.lisp:44
    (45 "Invalid data checksum (wanted ~8,'0x, got ~8,'0x)." :detail :check)              ; zlib.lisp:75
    (50 "Incorrect gzip magic bytes.")                                                    ; gzip.lisp:62
    (51 "Unrecognized compression method: ~x" :detail)                                    ; gzip.lisp:64
    (52 "Reserved flag bits are non-zero.")                                               ; gzip.lisp:66
    (53 "Inconsistent lengths for gzip extra fields.")                                    ; gzip.lisp:25
    (54 "Header checksum mismatch (required ~4,'0x, got ~4,'0x)." :detail :aux)           ; gzip.lisp:77
    (55 "Invalid data checksum (wanted ~8,'0x, got ~8,'0x)." :detail :check)              ; gzip.lisp:108
    (56 "Invalid data length check (wanted ~d, got ~d)." :detail :aux)                    ; gzip.lisp:112
    (60 "Incorrect bzip2 magic bytes: ~2,'0x ~2,'0x" :high-octet :low-octet)              ; bzip2.lisp:430
    (61 "Unrecognized bzip version: ~2,'0x" :detail)                                      ; bzip2.lisp:435
    (62 "Invalid bzip2 compression level: ~2,'0x" :detail)                                ; bzip2.lisp:437
    (63 "Unrecognized magic bytes on block boundary.")                                    ; bzip2.lisp:503
    (64 "Block declares no used symbols.")                                                ; bzip2.lisp:359
    (65 "Illegal tree count: ~d" :detail)                                                 ; bzip2.lisp:381
    (66 "Block defines no selectors.")                                                    ; bzip2.lisp:384
    (67 "Invalid selector: ~d" :detail)                                                   ; bzip2.lisp:307
    (68 "Code length goes out of bounds.")                                                ; bzip2.lisp:337
    (69 "Ran out of selectors before end of block.")                                      ; bzip2.lisp:250
    (70 "A/B run cannot possibly stay in bounds.")                                        ; bzip2.lisp:262
    (71 "A/B run expands beyond block size.")                                             ; bzip2.lisp:272
    (72 "Too many codes/expansions.")                                                     ; bzip2.lisp:283
    (73 "Invalid block CRC (expected ~8,'0x, got ~8,'0x)." :detail :check)                ; bzip2.lisp:520
    (74 "Incorrect stream CRC (expected ~8,'0x, got ~8,'0x)" :detail :check)              ; bzip2.lisp:493
    (80 "LZMA data doesn't begin with a zero octet.")                                     ; lzma.lisp:58
    (81 "Code starts out of range.")                                                      ; lzma.lisp:61
    (82 "Code goes out of range.")                                                        ; lzma.lisp:73, 188
    (83 "Match extends beyond dictionary size.")                                          ; lzma.lisp:605
    (84 "Match extends beyond available data.")                                           ; lzma.lisp:607
    (85 "Match extends beyond declared decompressed size.")                               ; lzma.lisp:609
    (86 "Decompressed data goes beyond declared size.")                                   ; lzma.lisp:718
    (87 "Didn't find mandatory EOF marker in LZMA data.")                                 ; lzma.lisp:643
    (88 "Illegal EOF marker in LZMA data.")                                               ; lzma.lisp:646, 730
    (89 "LZMA data continues beyond declared size.")                                      ; lzma.lisp:647
    (90 "Neither range decoder nor marker denote an end.")                                ; lzma.lisp:659
    (91 "Range decoder finishes in invalid state.")                                       ; lzma.lisp:774
    (92 "Decompressed data ends before declared size.")                                   ; lzma.lisp:894
    (93 "Property `pb' is out of bounds (0-4): ~d" :detail)                               ; lzma.lisp:842
    (100 "Property lc+lp out of bounds (0-4): ~d" :detail)                                ; lzma2.lisp:36
    (101 "Unrecognized LZMA2 dictionary size byte.")                                      ; lzma2.lisp:42
    (102 "Didn't get required dictionary reset.")                                         ; lzma2.lisp:98, 160
    (103 "Didn't get required property reset.")                                           ; lzma2.lisp:100
    (104 "Embedded LZMA data goes beyond LZMA2 block size.")                              ; lzma2.lisp:116
    (105 "Illegal LZMA2 control byte: ~2,'0x" :detail)                                    ; lzma2.lisp:149
    (106 "Embedded LZMA data is shorter than declared.")                                  ; lzma2.lisp:201
    (120 "Invalid XZ magic bytes.")                                                       ; xz.lisp:91
    (121 "Reserved header byte isn't zero.")                                              ; xz.lisp:93
    (122 "Incorrect header checksum.")                                                    ; xz.lisp:100
    (123 "Reserved header flag bits aren't zero.")                                        ; xz.lisp:102
    (124 "Unsupported checksum type: ~X" :detail)                                         ; xz.lisp:104
    (125 "Trailing garbage data after XZ stream.")                                        ; xz.lisp:133
    (126 "XZ padding doesn't come in multiples of 4 bytes.")                              ; xz.lisp:134
    (127 "Overlong multibyte integer encoding.")                                          ; xz.lisp:151
    (128 "Overlong multibyte integer.")                                                   ; xz.lisp:154
    (129 "Index padding contains non-zero bytes.")                                        ; xz.lisp:190
    (130 "Index disagrees with the observed number of blocks.")                           ; xz.lisp:196
    (131 "Index disagrees with the observed block sizes.")                                ; xz.lisp:203
    (132 "Incorrect index CRC (expected ~8,'0x, got ~8,'0x)." :detail :check)             ; xz.lisp:209
    (133 "Incorrect footer CRC (expected ~8,'0x, got ~8,'0x)." :detail :check)            ; xz.lisp:220
    (134 "Incorrect backwards size field.")                                               ; xz.lisp:225
    (135 "Inconsistent stream flags between header and footer.")                          ; xz.lisp:228
    (136 "Invalid footer magic bytes.")                                                   ; xz.lisp:230
    (137 "Reserved block header bits aren't zero.")                                       ; xz.lisp:267
    (138 "Block header is truncated.")                                                    ; xz.lisp:288
    (139 "Block header padding contains non-zero bytes.")                                 ; xz.lisp:291
    (140 "Embedded compressed data goes beyond declared XZ block size.")                  ; xz.lisp:309
    (141 "The Delta filter must not come last.")                                          ; xz.lisp:316
    (142 "Invalid property length for Delta filter: ~d" :detail)                          ; xz.lisp:318
    (143 "BCJ filters must not come last.")                                               ; xz.lisp:325
    (144 "Invalid property length for BCJ filter: ~d" :detail)                            ; xz.lisp:330
    (145 "Offset ~8,0x is not aligned to ~d" :detail :aux)                                ; xz.lisp:342
    (146 "The LZMA2 filter must always come last.")                                       ; xz.lisp:348
    (147 "Invalid property length for LZMA2 filter: ~d" :detail)                          ; xz.lisp:350
    (148 "Reserved LZMA property bits aren't zero.")                                      ; xz.lisp:352
    (149 "Unrecognized filter type: ~2,'0x" :detail)                                      ; xz.lisp:377
    (150 "Invalid block header CRC (expected ~8,'0x, got ~8,'0x)." :detail :check)        ; xz.lisp:404
    (151 "Total decompressed size exceeds 2^63 - 1 bytes.")                               ; xz.lisp:421
    (152 "XZ block is smaller than its declared size.")                                   ; xz.lisp:430
    (153 "Decompressed size doesn't match declared decompressed size.")                   ; xz.lisp:435
    (154 "Block padding doesn't consist of null bytes.")                                  ; xz.lisp:442
    (155 "Incorrect block checksum (expected ~v,'0x, got ~v,'0x)."
     :hex-digits :stored-check :hex-digits :check)                                        ; xz.lisp:452-454
    (156 "XZ stream is longer than 2^63 - 1 bytes.")                                      ; xz.lisp:87
    (170 "Trailing data detected.")))                                                     ; interface.lisp:130

(defstruct (failure (:constructor make-failure (status detail aux check check-size stored-check)))
  "Where a decoded stream stopped: the status and the arguments of the reference's message for it."
  status detail aux check check-size stored-check)

(defun signal-failure (failure)
  "Signals what the reference signals for the status, with the reference's message and arguments."
  (let ((status (failure-status failure))
        (detail (failure-detail failure)))
    (cond ((= status 1) (semz.decompress::%eof))                         ; common.lisp:56-57
          ((= status 44)                                                 ; zlib.lisp:47-48
           (error 'semz.decompress:unrecognized-zlib-dictionary :checksum detail))
          ((>= status 200)                                               ; engine-level, no counterpart in the reference
           (error 'engine-error :message (semz-strerror status)))
          (t
           (let ((entry (assoc status *messages*)))
             (if (null entry)
                 (semz.decompress::die "~a" (semz-strerror status))      ; common.lisp:53-54
                 (apply #'semz.decompress::die (second entry)
                        (loop :for source :in (cddr entry)
                              :collect (ecase source
                                         (:detail detail)
                                         (:aux (failure-aux failure))
                                         (:check (failure-check failure))
                                         (:high-octet (ldb (byte 8 8) detail))
                                         (:low-octet (ldb (byte 8 0) detail))
                                         (:hex-digits (* 2 (failure-check-size failure)))
                                         (:stored-check (failure-stored-check failure)))))))))))

(defparameter *header-statuses*
  '(40 41 42 43 44          ; zlib.lisp:30-48
    50 51 52 53 54          ; gzip.lisp:24-25, 60-78
    60 61 62                ; bzip2.lisp:429-437
    93 101                  ; lzma.lisp:841-842, lzma2.lisp:42
    120 121 122 123 124)    ; xz.lisp:90-104
  "Statuses the reference signals while it reads a member's header, i.e. inside `byte-source->decompression-state'.")

;;;; Header plists of the first member (semz_stream_info.first_header, INTEGRATION.md section 1)

(defun header-plist (format info buffer base)
  (flet ((word (i) (cffi:mem-aref (cffi:foreign-slot-pointer info '(:struct semz-stream-info) 'first-header) :uint32 i))
         (latin-1 (off len)
           (map 'string #'code-char (subseq buffer (+ base off) (+ base off len)))))
    (ecase format
      ((:deflate :raw-lzma :raw-lzma2) nil)
      (:zlib                                                             ; zlib.lisp:57
       (list :window-size (word 0) :level (word 1)
             :dictionary (if (zerop (word 2)) nil (word 3))))
      (:gzip                                                             ; gzip.lisp:81-87
       (list :textp (logbitp 0 (word 0))
             :extra-fields (if (= (word 2) #xFFFFFFFF)
                               '()
                               (semz.decompress::parse-gzip-extra-fields  ; gzip.lisp:23-29
                                (subseq buffer (+ base (word 2)) (+ base (word 2) (word 3)))))
             :filename (if (= (word 4) #xFFFFFFFF) nil (latin-1 (word 4) (word 5)))
             :comment (if (= (word 6) #xFFFFFFFF) nil (latin-1 (word 6) (word 7)))
             :modification-time (word 1)
             :extra-flags (ldb (byte 8 8) (word 0))
             :operating-system (ldb (byte 8 16) (word 0))))
      (:bzip2 (list :block-size (* 100000 (word 0))))                    ; bzip2.lisp:438-449
      (:xz (list :checksum-type (word 0)))                               ; xz.lisp:111-112
      (:lzma2 (list :window-size (word 0)))                              ; lzma2.lisp:226-227
      (:lzma                                                             ; lzma.lisp:860-865
       (list :lc (word 0) :lp (word 1) :pb (word 2) :window-size (word 3)
             :decompressed-size (if (zerop (word 4)) nil (logior (word 5) (ash (word 6) 32))))))))

;;;; The state

(defclass gpu-batch-state ()
  ((%format :initarg :format :reader gbs-format)
   (%source :initarg :source :reader gbs-source)        ; the fully buffered buffer-stream (io.lisp:16-23)
   (%base :initarg :base :reader gbs-base)              ; its `start' when the input was handed to the engine
   (%consumed :initarg :consumed :reader gbs-consumed)  ; octets of input the whole decode consumed
   (%handle :initarg :handle :accessor gbs-handle)      ; semz_decoded *, nil once released
   (%members :initarg :members :reader gbs-members)     ; simple-vector of (in-end . out-end), one per decoded member
   (%index :initform 0 :accessor gbs-index)             ; the member being handed out
   (%offset :initform 0 :accessor gbs-offset)           ; octets of output handed out so far
   (%out-len :initarg :out-len :reader gbs-out-len)     ; octets the whole decode produced: the members, and what the
                                                        ; member that failed had delivered before its bad data
   (%pending :initarg :pending :accessor gbs-pending)   ; `failure' of the member after the last good one, or nil
   (%chunk :initform semz.decompress::+dummy-buffer+ :accessor gbs-chunk))
  (:documentation "All members of one input, decoded by the engine and waiting in device memory."))

(defun release-decoded (state)
  (when (gbs-handle state)
    #+sbcl (sb-ext:cancel-finalization state)
    (semz-stream-release (gbs-handle state))
    (setf (gbs-handle state) nil)))

(defun advance-source (state input-octets)
  "Puts the byte source behind `input-octets' octets of the input: `buffer-stream-check-eof' (interface.lisp:110-112)
and callers that go on reading after a member look at it."
  (setf (semz.decompress::buffer-stream-start (gbs-source state))
        (min (semz.decompress::buffer-stream-end (gbs-source state))
             (+ (gbs-base state) input-octets))))

(defun fetch-chunk (state n)
  "Octets [offset, offset + n) of the decoded stream into the state's chunk vector, straight from the device."
  (when (< (length (gbs-chunk state)) n)
    (setf (gbs-chunk state)
          (make-array (min *chunk-size* (max n 65536)) :element-type 'semz.decompress::octet)))
  (when (plusp n)
    (cffi:with-pointer-to-vector-data (dst (gbs-chunk state))            ; pins the vector for the copy
      (check-rc (semz-stream-read (gbs-handle state) (gbs-offset state) dst n) "semz_stream_read"))))

(defmethod semz.decompress::next-decompressed-chunk ((state gpu-batch-state))
  (let ((members (gbs-members state))
        (k (gbs-index state)))
    (cond
      ((< k (length members))
       (destructuring-bind (in-end . out-end) (svref members k)
         (let ((n (min *chunk-size* (- out-end (gbs-offset state)))))
           (fetch-chunk state n)
           (incf (gbs-offset state) n)
           (let ((finalp (= (gbs-offset state) out-end)))
             (when finalp
               (advance-source state in-end)
               (when (and (= (1+ k) (length members)) (null (gbs-pending state)))
                 (release-decoded state)))
             (values (gbs-chunk state) 0 n finalp)))))
      ;; The member that failed: the reference hands out every chunk it decodes before it reaches the bad data
      ;; (interface.lisp:95-108), and the engine has those octets behind the last complete member.
      ((and (gbs-pending state) (< (gbs-offset state) (gbs-out-len state)))
       (let ((n (min *chunk-size* (- (gbs-out-len state) (gbs-offset state)))))
         (fetch-chunk state n)
         (incf (gbs-offset state) n)
         (values (gbs-chunk state) 0 n nil)))
      ((gbs-pending state)
       (release-decoded state)
       (signal-failure (gbs-pending state)))
      (t (values semz.decompress::+dummy-buffer+ 0 0 t)))))

(defmethod semz.decompress::make-reset-state ((state gpu-batch-state))
  (cond
    ;; single-member formats: the caller says "Trailing data detected." (interface.lisp:125-130)
    ((not (member (gbs-format state) *multi-member-formats*)) nil)
    (t
     (incf (gbs-index state))
     (when (and (>= (gbs-index state) (length (gbs-members state)))
                (null (gbs-pending state)))
       ;; nothing but what the engine already accepted is left (xz stream padding, xz.lisp:118-134)
       (advance-source state (gbs-consumed state)))
     ;; the header of a later member is ignored by the caller (interface.lisp:118-122)
     (values state nil))))

;;;; Dispatch

(defvar *cpu-path* nil
  "True while one of the reference's own methods runs: what it calls in turn (gzip -> :deflate, xz -> :raw-lzma2 ->
:raw-lzma, `make-reset-state' -> the next member's header) belongs to the reference's decoder.")

(defmethod semz.decompress::make-reset-state :around (old-state)
  (if (typep old-state 'gpu-batch-state)
      (call-next-method)
      (let ((*cpu-path* t))
        (call-next-method))))

(defun engine-options (format args)
  "The (name . value) options of semz_set_option that `args' ask for, or :unsupported when the reference has to
decode this call.  check-types as in the reference's methods (lzma.lisp:818-825, lzma2.lisp:212)."
  (when (loop :for key :in args :by #'cddr :thereis (not (keywordp key)))
    (return-from engine-options :unsupported))           ; the reference's internal buffer-reuse options
  (ecase format
    ((:gzip :bzip2 :xz :lzma :lzma2) '())
    (:deflate (if (or (getf args :prefix)                                ; deflate.lisp:412-425
                      (/= (getf args :window-size 32768) 32768))
                  :unsupported
                  '()))
    (:zlib (if (getf args :dictionary) :unsupported '()))                ; zlib.lisp:37-56
    (:raw-lzma2
     (let ((window-size (getf args :window-size)))
       (check-type window-size unsigned-byte)
       (list (cons "window-size" (min window-size (1- (expt 2 32)))))))
    (:raw-lzma
     (let ((lc (getf args :lc)) (lp (getf args :lp)) (pb (getf args :pb))
           (window-size (getf args :window-size))
           (decompressed-size (getf args :decompressed-size))
           (eof-mode (or (getf args :eof-mode) :maybe)))
       (check-type lc (integer 0 8))
       (check-type lp (integer 0 4))
       (check-type pb (integer 0 4))
       (check-type window-size unsigned-byte)
       (check-type decompressed-size (or null unsigned-byte))
       (check-type eof-mode (member :always :never :maybe))
       (when (and decompressed-size (>= decompressed-size (expt 2 62)))
         (return-from engine-options :unsupported))
       (list (cons "lc" lc) (cons "lp" lp) (cons "pb" pb)
             (cons "window-size" (min window-size (1- (expt 2 32))))
             (cons "decompressed-size" (or decompressed-size -1))
             (cons "eof-mode" (ecase eof-mode (:maybe 0) (:always 1) (:never 2))))))))

(defun batch-source-p (byte-source)
  "True for the zero-copy wrapper of a `buffer'-typed vector (array->buffer-stream, io.lisp:30-31): a plain
buffer-stream that already holds data.  The copying wrapper of other arrays (io.lisp:32-40) and the wrapper of a
stream (io.lisp:42-57) start out empty, counted sources are views of another source (io.lisp:218-225)."
  (and (semz.decompress::buffer-stream-p byte-source)
       (not (semz.decompress::counted-buffer-stream-p byte-source))
       (>= (- (semz.decompress::buffer-stream-end byte-source)
              (semz.decompress::buffer-stream-start byte-source))
           (max 1 *minimum-input-size*))))

(defun make-gpu-state (format byte-source options)
  "One semz_stream_decode call over the rest of `byte-source'.  Returns the state and the header plist."
  (open-engine)
  (loop :for (name . value) :in options
        :do (check-rc (semz-set-option *engine* name value) "semz_set_option"))
  (let* ((buffer (semz.decompress::buffer-stream-buffer byte-source))
         (base (semz.decompress::buffer-stream-start byte-source))
         (end (semz.decompress::buffer-stream-end byte-source))
         ;; multi-member formats are decoded to the end: which members the caller wants is decided later, by
         ;; whether it calls `make-reset-state' (all-members-p never reaches this function, interface.lisp:85)
         (all-members (if (member format *multi-member-formats*) 1 0)))
    (cffi:with-foreign-objects ((decoded :pointer)
                                (res '(:struct semz-result))
                                (info '(:struct semz-stream-info))
                                (members :pointer)
                                (nmembers :size))
      (cffi:with-pointer-to-vector-data (in buffer)                      ; pins the caller's vector for the call
        (check-rc (semz-stream-decode *engine* (getf *format-codes* format)
                                      (cffi:inc-pointer in base) (- end base) all-members
                                      decoded res info)
                  "semz_stream_decode"))
      (let ((handle (cffi:mem-ref decoded :pointer)))
        (check-rc (semz-stream-members handle members nmembers) "semz_stream_members")
        (let* ((n (cffi:mem-ref nmembers :size))
               (table (make-array n))
               (status (cffi:foreign-slot-value res '(:struct semz-result) 'status))
               (detail (cffi:foreign-slot-value res '(:struct semz-result) 'detail))
               (consumed (ash (cffi:foreign-slot-value res '(:struct semz-result) 'in-end-bit) -3)))
          (dotimes (i n)
            (let ((m (cffi:mem-aptr (cffi:mem-ref members :pointer) '(:struct semz-member) i)))
              (setf (svref table i)
                    (cons (cffi:foreign-slot-value m '(:struct semz-member) 'in-end)
                          (cffi:foreign-slot-value m '(:struct semz-member) 'out-end)))))
          (let ((failure
                  (if (zerop status)
                      nil
                      (let* ((check-len (cffi:foreign-slot-value res '(:struct semz-result) 'check-len))
                             (check-ptr (cffi:foreign-slot-pointer res '(:struct semz-result) 'check))
                             ;; the xz block's check field ends where the decode stopped (xz.lisp:444-450)
                             (stored-end (+ base consumed)))
                        (make-failure status detail
                                      (cffi:foreign-slot-value res '(:struct semz-result) 'aux)
                                      (loop :for i :below check-len   ; little-endian, as the reference compares it
                                            :sum (ash (cffi:mem-aref check-ptr :uint8 i) (* 8 i)))
                                      check-len
                                      (if (and (= status 155) (<= (+ base check-len) stored-end end))
                                          (loop :for i :below check-len
                                                :sum (ash (aref buffer (+ (- stored-end check-len) i)) (* 8 i)))
                                          0))))))
            ;; a header the reference rejects is rejected here, where the reference rejects it
            (when (and failure (zerop n) (member status *header-statuses*))
              (semz-stream-release handle)
              (signal-failure failure))
            (let ((state (make-instance 'gpu-batch-state
                                        :format format :source byte-source :base base :consumed consumed
                                        :handle handle :members table
                                        :out-len (cffi:foreign-slot-value res '(:struct semz-result) 'out-len)
                                        :pending failure)))
              ;; a caller that stops reading early (first member only, an abandoned stream) never reaches the
              ;; release in `next-decompressed-chunk': the collector returns the device memory then
              ;; (after `close-engine' too: semz_close cuts the decoded streams of its context loose, and releasing one
              ;; of those frees host memory only)
              #+sbcl (sb-ext:finalize state (lambda () (semz-stream-release handle)) :dont-save t)
              (values state (header-plist format info buffer base)))))))))

(macrolet ((define-dispatch (&rest formats)
             `(progn
                ,@(loop :for format :in formats
                        :collect
                        `(defmethod semz.decompress::byte-source->decompression-state :around
                             ((format (eql ,format)) byte-source &rest args &key &allow-other-keys)
                           (let ((options (if (and *enabled* (not *cpu-path*) (batch-source-p byte-source))
                                              (engine-options format args)
                                              :unsupported)))
                             (if (eq options :unsupported)
                                 (let ((*cpu-path* t))
                                   (call-next-method))
                                 (make-gpu-state format byte-source options))))))))
  (define-dispatch :deflate :zlib :gzip :bzip2 :raw-lzma :lzma :raw-lzma2 :lzma2 :xz))
