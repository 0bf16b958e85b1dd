/*
 * glsl_eval.js — evaluator for the GLSL ES 1.00 subset used by the reference's shaders.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/make_golden.js plugs it into the Node import of the
 * reference so that the reference's OWN host JavaScript and OWN shader strings (read from
 * /root/reference at generation time, never written into this repository) run in software.
 * The numeric outputs are committed as tests/golden/swgl_*; they check that the CPU
 * restatement (oracle/pic_oracle.c, .js) transcribes the shader text faithfully: operand
 * order, swizzles, texture bindings, constants, pass order, blending.  They do NOT pin what a
 * real GPU computes: precision of built-ins, fused multiply-adds and rasteriser snapping are
 * implementation-defined in GLSL ES 1.00 and are fixed here by convention:
 *   - every arithmetic operation rounds to float32 (Math.fround), no contraction;
 *   - dot(a,b) sums left to right, length(v) = sqrt(dot(v,v)), sqrt and divide are IEEE,
 *     cos is Math.cos rounded to float32;
 *   - texture2D is NEAREST + CLAMP_TO_EDGE: texel = clamp(floor(u*W), 0, W-1), NaN -> 0.
 * A SECOND convention set, setConvention('gpu'), makes the other common choices of a GPU's GLSL compiler
 * — a*b + c and a*b - c contracted into one fused multiply-add, dot() as a chain of them, x / y as
 * x * (1 / y) with a rounded reciprocal, the viewport transform of a point in float32 (swgl.js) — so that
 * running the reference under both gives an honest error bar on "matches the reference" for a real browser
 * GPU (tests/golden/conventions.json, tests/test_oracle_conventions.py).  Neither set is "the" GPU.
 * Since round 4 the first set has been checked against a real GLSL compiler: the reference run under Chromium's WebGL
 * (oracle/make_golden_webgl.py, tests/golden/webgl_*) gives the same numbers wherever no transcendental and no division by a
 * varying is involved (tests/test_oracle_webgl.py::test_software_evaluator_against_the_real_compiler); those fixtures, not
 * these, are what pins the oracle's arithmetic now.
 *
 * Supported: precision / uniform / varying / attribute declarations; void main(); float,
 * vec2-4 locals; = and += ; if / else; for (float i = a; i < b; i++); ternaries; + - * / ;
 * comparisons; || && ; swizzles; constructors; texture2D, sqrt, length, dot, cross, cos,
 * sign, abs, min, max, floor, fract, mod; gl_FragColor, gl_Position, gl_PointSize, gl_PointCoord.
 */
'use strict';

const f = Math.fround;

let CONVENTION = 'ieee';
function setConvention(name) {
    if (name !== 'ieee' && name !== 'gpu') throw new Error('glsl: unknown convention ' + name);
    CONVENTION = name;
}
// a*b + c with one rounding: the product of two float32 is exact in a double; the sum is rounded to double
// and then to float32 (the double rounding differs from a true fma in rare ties only)
function fma(a, b, c) { return f(a * b + c); }

// ------------------------------------------------------------------ tokenizer
function tokenize(src) {
    const toks = [];
    const re = /\s+|\/\/[^\n]*|\/\*[\s\S]*?\*\/|(\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+|\d+)|([A-Za-z_]\w*)|(\+\+|--|\+=|-=|\*=|\/=|<=|>=|==|!=|&&|\|\||[-+*\/=<>!?:;,.(){}\[\]])/y;
    let pos = 0;
    while (pos < src.length) {
        re.lastIndex = pos;
        const m = re.exec(src);
        if (!m) throw new Error('glsl: cannot tokenize at ' + src.slice(pos, pos + 30));
        pos = re.lastIndex;
        if (m[1] !== undefined) toks.push({ t: 'num', v: Number(m[1]) });
        else if (m[2] !== undefined) toks.push({ t: 'id', v: m[2] });
        else if (m[3] !== undefined) toks.push({ t: 'op', v: m[3] });
    }
    return toks;
}

// ------------------------------------------------------------------ parser
const TYPES = { float: 1, vec2: 2, vec3: 3, vec4: 4, int: 1, bool: 1 };

function parse(src) {
    const toks = tokenize(src);
    let p = 0;
    const peek = function (v) { return p < toks.length && toks[p].v === v && toks[p].t !== 'num'; };
    const next = function () { return toks[p++]; };
    const expect = function (v) { if (!peek(v)) throw new Error('glsl: expected ' + v + ' near token ' + p + ' (' + (toks[p] && toks[p].v) + ')'); p++; };
    const decls = { uniform: {}, varying: {}, attribute: {} };

    function primary() {
        const t = next();
        if (t.t === 'num') return { k: 'num', v: f(t.v) };
        if (t.t === 'op' && t.v === '(') { const e = expr(); expect(')'); return postfix(e); }
        if (t.t === 'id') {
            if (peek('(')) {
                p++;
                const args = [];
                if (!peek(')')) { do { args.push(assignExpr()); } while (peek(',') && p++); }
                expect(')');
                return postfix({ k: 'call', name: t.v, args: args });
            }
            return postfix({ k: 'var', name: t.v });
        }
        throw new Error('glsl: unexpected token ' + t.v);
    }
    function postfix(e) {
        while (peek('.')) { p++; e = { k: 'swz', e: e, s: next().v }; }
        if (peek('++')) { p++; e = { k: 'postinc', e: e }; }
        return e;
    }
    function unary() {
        if (peek('-')) { p++; return { k: 'neg', e: unary() }; }
        if (peek('+')) { p++; return unary(); }
        if (peek('!')) { p++; return { k: 'not', e: unary() }; }
        return primary();
    }
    function bin(sub, ops) {
        return function () {
            let e = sub();
            for (;;) {
                let hit = null;
                for (const o of ops) if (peek(o)) hit = o;
                if (!hit) return e;
                p++;
                e = { k: 'bin', op: hit, a: e, b: sub() };
            }
        };
    }
    const mul = bin(unary, ['*', '/']);
    const add = bin(mul, ['+', '-']);
    const rel = bin(add, ['<', '>', '<=', '>=']);
    const eq = bin(rel, ['==', '!=']);
    const land = bin(eq, ['&&']);
    const lor = bin(land, ['||']);
    function ternary() {
        const c = lor();
        if (!peek('?')) return c;
        p++;
        const a = assignExpr();
        expect(':');
        const b = assignExpr();
        return { k: 'tern', c: c, a: a, b: b };
    }
    function assignExpr() {
        const l = ternary();
        for (const o of ['=', '+=', '-=', '*=', '/=']) {
            if (peek(o)) { p++; return { k: 'assign', op: o, l: l, r: assignExpr() }; }
        }
        return l;
    }
    function expr() { return assignExpr(); }

    function statement() {
        if (peek('{')) { p++; const body = []; while (!peek('}')) body.push(statement()); p++; return { k: 'block', body: body }; }
        if (peek('if')) {
            p++; expect('('); const c = expr(); expect(')');
            const a = statement();
            let b = null;
            if (peek('else')) { p++; b = statement(); }
            return { k: 'if', c: c, a: a, b: b };
        }
        if (peek('for')) {
            p++; expect('(');
            const init = statement();          // consumes its ';'
            const cond = expr(); expect(';');
            const step = expr(); expect(')');
            return { k: 'for', init: init, cond: cond, step: step, body: statement() };
        }
        if (toks[p].t === 'id' && TYPES[toks[p].v] && toks[p + 1] && toks[p + 1].t === 'id') {
            const type = next().v, name = next().v;
            let init = null;
            if (peek('=')) { p++; init = expr(); }
            expect(';');
            return { k: 'decl', type: type, name: name, init: init };
        }
        const e = expr();
        expect(';');
        return { k: 'expr', e: e };
    }

    let main = null;
    while (p < toks.length) {
        if (peek('precision')) { p += 3; expect(';'); continue; }
        if (peek('uniform') || peek('varying') || peek('attribute')) {
            const q = next().v, type = next().v, name = next().v;
            expect(';');
            decls[q][name] = type;
            continue;
        }
        if (peek('void')) {
            p++;
            if (next().v !== 'main') throw new Error('glsl: only main() is supported');
            expect('('); expect(')');
            main = statement();
            continue;
        }
        throw new Error('glsl: unsupported top-level token ' + toks[p].v);
    }
    return { decls: decls, main: main };
}

// ------------------------------------------------------------------ evaluator
const isVec = Array.isArray;
const SWZ = { x: 0, y: 1, z: 2, w: 3, r: 0, g: 1, b: 2, a: 3, s: 0, t: 1, p: 2, q: 3 };

function map2(a, b, fn) {
    if (isVec(a) && isVec(b)) { if (a.length !== b.length) throw new Error('glsl: vector size mismatch'); return a.map(function (x, i) { return fn(x, b[i]); }); }
    if (isVec(a)) return a.map(function (x) { return fn(x, b); });
    if (isVec(b)) return b.map(function (y) { return fn(a, y); });
    return fn(a, b);
}
const OPS = {
    '+': function (x, y) { return f(x + y); }, '-': function (x, y) { return f(x - y); },
    '*': function (x, y) { return f(x * y); },
    '/': function (x, y) { return CONVENTION === 'gpu' ? f(x * f(1 / y)) : f(x / y); },
};
function dot(a, b) {
    let s = f(a[0] * b[0]);
    for (let i = 1; i < a.length; i++) s = CONVENTION === 'gpu' ? fma(a[i], b[i], s) : f(s + f(a[i] * b[i]));
    return s;
}
// element-wise a*b + sign*c (or c + sign*a*b) over scalars and vectors
function map3(a, b, c, fn) {
    const n = Math.max(isVec(a) ? a.length : 0, isVec(b) ? b.length : 0, isVec(c) ? c.length : 0);
    if (n === 0) return fn(a, b, c);
    const at = function (v, i) { return isVec(v) ? v[i] : v; };
    const out = [];
    for (let i = 0; i < n; i++) out.push(fn(at(a, i), at(b, i), at(c, i)));
    return out;
}

function sample(tex, uv) {
    const W = tex.width, H = tex.height;
    const idx = function (u, n) { const t = f(u * n); if (!(t >= 0)) return 0; if (t >= n) return n - 1; return Math.floor(t); };
    const o = 4 * (idx(uv[0], W) + W * idx(uv[1], H));
    const a = tex.array;
    return [a[o], a[o + 1], a[o + 2], a[o + 3]];
}

const BUILTINS = {
    sqrt: function (a) { return isVec(a[0]) ? a[0].map(function (x) { return f(Math.sqrt(x)); }) : f(Math.sqrt(a[0])); },
    cos: function (a) { return f(Math.cos(a[0])); },
    floor: function (a) { return isVec(a[0]) ? a[0].map(Math.floor) : Math.floor(a[0]); },
    fract: function (a) { const g = function (x) { return f(x - Math.floor(x)); }; return isVec(a[0]) ? a[0].map(g) : g(a[0]); },
    // mod(x, y) = x - y * floor(x / y), each step rounded (GLSL ES 1.00 section 8.3)
    mod: function (a) { return map2(a[0], a[1], function (x, y) { return f(x - f(y * Math.floor(f(x / y)))); }); },
    abs: function (a) { return isVec(a[0]) ? a[0].map(Math.abs) : Math.abs(a[0]); },
    sign: function (a) { const s = function (x) { return x > 0 ? 1 : (x < 0 ? -1 : 0); }; return isVec(a[0]) ? a[0].map(s) : s(a[0]); },
    min: function (a) { return map2(a[0], a[1], function (x, y) { return y < x ? y : x; }); },
    max: function (a) { return map2(a[0], a[1], function (x, y) { return x < y ? y : x; }); },
    dot: function (a) { return dot(a[0], a[1]); },
    length: function (a) { return isVec(a[0]) ? f(Math.sqrt(dot(a[0], a[0]))) : Math.abs(a[0]); },
    cross: function (a) {
        const u = a[0], v = a[1];
        return [f(f(u[1] * v[2]) - f(u[2] * v[1])), f(f(u[2] * v[0]) - f(u[0] * v[2])), f(f(u[0] * v[1]) - f(u[1] * v[0]))];
    },
    texture2D: function (a) { return sample(a[0], a[1]); },
};
function construct(n, args) {
    const flat = [];
    args.forEach(function (v) { if (isVec(v)) v.forEach(function (x) { flat.push(x); }); else flat.push(f(v)); });
    if (flat.length === 1 && n > 1) { const o = []; for (let i = 0; i < n; i++) o.push(flat[0]); return o; }
    if (flat.length < n) throw new Error('glsl: too few constructor arguments');
    return n === 1 ? flat[0] : flat.slice(0, n);
}

function run(ast, env) {
    // env: name -> value (number | array | texture object)
    function lvalueSet(node, val) {
        if (node.k === 'var') { env[node.name] = isVec(val) ? val.slice() : val; return; }
        if (node.k === 'swz') {
            const base = ev(node.e).slice();
            const comps = node.s.split('');
            if (comps.length === 1) base[SWZ[comps[0]]] = val; else comps.forEach(function (c, i) { base[SWZ[c]] = val[i]; });
            lvalueSet(node.e, base);
            return;
        }
        throw new Error('glsl: bad assignment target');
    }
    function ev(n) {
        switch (n.k) {
        case 'num': return n.v;
        case 'var': {
            if (!(n.name in env)) {
                // an undefined gl_FragColor read by `+=` (quirk Q7) starts at 0
                if (n.name === 'gl_FragColor') return [0, 0, 0, 0];
                throw new Error('glsl: undefined variable ' + n.name);
            }
            return env[n.name];
        }
        case 'neg': { const v = ev(n.e); return isVec(v) ? v.map(function (x) { return -x; }) : -v; }
        case 'not': return !ev(n.e);
        case 'swz': {
            const v = ev(n.e);
            const comps = n.s.split('');
            if (comps.length === 1) return v[SWZ[comps[0]]];
            return comps.map(function (c) { return v[SWZ[c]]; });
        }
        case 'bin': {
            if (n.op === '||') return ev(n.a) || ev(n.b);
            if (n.op === '&&') return ev(n.a) && ev(n.b);
            if (CONVENTION === 'gpu' && (n.op === '+' || n.op === '-')) {
                // contraction: a product feeding an add or a subtract becomes one fused multiply-add
                const isMul = function (x) { return x.k === 'bin' && x.op === '*'; };
                if (isMul(n.a)) {
                    const c = ev(n.b);
                    return map3(ev(n.a.a), ev(n.a.b), c, function (x, y, z) { return fma(x, y, n.op === '+' ? z : -z); });
                }
                if (isMul(n.b)) {
                    const c = ev(n.a);
                    return map3(ev(n.b.a), ev(n.b.b), c, function (x, y, z) { return fma(n.op === '+' ? x : -x, y, z); });
                }
            }
            const a = ev(n.a), b = ev(n.b);
            switch (n.op) {
            case '<': return a < b; case '>': return a > b; case '<=': return a <= b; case '>=': return a >= b;
            case '==': return a === b; case '!=': return a !== b;
            default: return map2(a, b, OPS[n.op]);
            }
        }
        case 'tern': return ev(n.c) ? ev(n.a) : ev(n.b);
        case 'call': {
            if (TYPES[n.name]) return construct(TYPES[n.name], n.args.map(ev));
            const fn = BUILTINS[n.name];
            if (!fn) throw new Error('glsl: unsupported function ' + n.name);
            return fn(n.args.map(ev));
        }
        case 'assign': {
            let v = ev(n.r);
            if (n.op !== '=') v = map2(ev(n.l), v, OPS[n.op[0]]);
            lvalueSet(n.l, v);
            return v;
        }
        case 'postinc': { const v = ev(n.e); lvalueSet(n.e, f(v + 1)); return v; }
        default: throw new Error('glsl: bad expression node ' + n.k);
        }
    }
    function ex(s) {
        switch (s.k) {
        case 'block': s.body.forEach(ex); return;
        case 'decl': env[s.name] = s.init ? (function (v) { return isVec(v) ? v.slice() : v; })(ev(s.init)) : (TYPES[s.type] > 1 ? new Array(TYPES[s.type]).fill(0) : 0); return;
        case 'expr': ev(s.e); return;
        case 'if': if (ev(s.c)) ex(s.a); else if (s.b) ex(s.b); return;
        case 'for': for (ex(s.init); ev(s.cond); ev(s.step)) ex(s.body); return;
        default: throw new Error('glsl: bad statement ' + s.k);
        }
    }
    ex(ast.main);
    return env;
}

module.exports = { parse: parse, run: run, fround: f, setConvention: setConvention, convention: function () { return CONVENTION; } };
