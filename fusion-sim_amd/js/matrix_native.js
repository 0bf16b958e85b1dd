/*
 * matrix_native.js — drop-in for the reference's `matrix_webgl` module on MI355X.
 *
 *   const matrix = require('./matrix_native.js');
 *   const eq = matrix.makeSORIterative({ n_power: 3, relaxation: 0.9 });   // matrix_webgl.js:35
 *   eq.set_matrix(A).set_b(b).init_vector(x0);                            // :456, :479, :500
 *   const r = eq.solve({ tolerance: 1e-6, substep: 2, max_iterations: 50 }); // :566
 *   // r = { correlation, diff, iterations, result: Float32Array(vec_length) }
 *
 * Same factory name, method names, argument meaning, chaining and synchronous
 * `throw new Error(".prop <- ...")` as the reference object; the arithmetic runs in HIP kernels
 * behind libfusionpic.so (include/fusionsor.h), bit-identical to the reference's shader passes
 * evaluated in IEEE float32.  No JavaScript compute path exists: without the addon or a gfx950
 * device the factory throws.
 *
 * Differences a caller can see:
 *   - `spec.webgl` is accepted and ignored (there is no GL context); `out.canvas` does not exist;
 *   - matrices and vectors may also be Float32Array / Float64Array (row-major), which is the
 *     only practical form for n_power >= 5 (vec_length^2 >= 1.6e7 elements);
 *   - x_result_tex() returns { read(out?) -> Float32Array } instead of a WebGL frame buffer;
 *   - solve() does not print R, C and every iterate to the console (matrix_webgl.js:592-597, :676);
 *   - spec.compat (default true) keeps the reference's row permutation of the update
 *     (matrix_webgl.js:389-424: element e receives matrix row (2X + c%2) + 2vh(2Y + c/2)), with
 *     which the iteration does not converge to the solution of A x = b once n_power > 0;
 *     compat: false uses row e.  n_power = 0 throws as in the reference (programResult cannot
 *     be linked against sum_buffers[-1]).
 */
'use strict';
const empic = require('./empic_native.js');

function flatten(a, rows, cols, what) {
    if (a instanceof Float32Array || a instanceof Float64Array) return a;
    if (!Array.isArray(a)) throw new Error(what + ' must be an array or a Float32Array/Float64Array');
    if (cols === 0) return Float64Array.from(a);
    const out = new Float64Array(rows * cols);
    for (let r = 0; r < rows; r++) for (let c = 0; c < cols; c++) out[c + cols * r] = a[r][c];
    return out;
}

exports.makeSORIterative = function (spec) {
    empic.validate_object(spec, {
        n_power: 'number',
        relaxation: [, 'number'],
        webgl: [, 'object'],
        device: [, 'number'],
        compat: [, 'boolean'],
    });
    const native = empic._addon();
    const h = native.sorCreate(spec.n_power, spec.relaxation || 0, spec.device || 0, spec.compat === false ? 1 : 0);
    const dims = native.sorDims(h);
    const out = { vec_length: dims[0], vec_height: dims[1] };
    const L = out.vec_length;

    out.set_matrix = function (matrix) { native.sorSet(h, 0, flatten(matrix, L, L, 'matrix')); return out; };
    out.set_b = function (b) { native.sorSet(h, 1, flatten(b, L, 0, 'b')); return out; };
    out.init_vector = function (vector) { native.sorSet(h, 2, flatten(vector, L, 0, 'vector')); return out; };
    // x_guess <- x_result; x_result <- R x_guess + C.  `target` is accepted for signature
    // compatibility; the product always lands in x_result (matrix_webgl.js:535-558)
    out.mv_product = function (target) { native.sorIterate(h, 1); return out; };
    out.solve = function (params) {
        empic.validate_object(params, { tolerance: 'number', substep: [, 'number'], max_iterations: [, 'number'] });
        const result = new Float32Array(L);
        const has_max = typeof params.max_iterations === 'number';
        const r = native.sorSolve(h, params.tolerance, params.substep || 0, has_max ? 1 : 0, has_max ? params.max_iterations : 0, result);
        return { correlation: r[0], diff: r[1], iterations: r[2], result: result };
    };
    out.x_result_tex = function () {
        return { width: out.vec_height, height: out.vec_height,
            read: function (arr) { arr = arr || new Float32Array(L); native.sorRead(h, 0, arr); return arr; } };
    };
    // ---- extensions
    out.readVector = function (which, arr) { arr = arr || new Float32Array(L); native.sorRead(h, which, arr); return arr; };
    out.readIterationMatrix = function (arr) { arr = arr || new Float32Array(L * L); native.sorRead(h, -1, arr); return arr; };
    out.iterate = function (n) { native.sorIterate(h, n); return out; };
    out.sync = function () { native.sorSync(h); return out; };
    out.destroy = function () { native.sorDestroy(h); };
    return out;
};
